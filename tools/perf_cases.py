"""Geometries the dev tools time (tools/perf_matrix.py, tools/stamps.py): one per kernel tier and the sub-roofline corners."""
from srsran_ce_pytorch_amd import synth as S
H, CS = S.hop_spec, S.case_spec
BOTH = [S.TYPE1_CDM0, S.TYPE1_CDM1]
CASES = [
    ("L1 2dmrs filter (register path)", S.bench_case("filter", 1), "linear"),
    ("L1 2dmrs none", S.bench_case("none", 1), "linear"),
    ("L1 2dmrs mean", S.bench_case("mean", 1), "linear"),
    ("L2 2dmrs filter (register path)", S.bench_case("filter", 2), "linear"),
    ("L4 2dmrs filter (generic path)", S.bench_case("filter", 4), "linear"),
    ("L1 3dmrs filter (register path)", CS("d3", 273, [H([2, 7, 11], 0, 273)]), "linear"),
    ("L1 2 hops x 2dmrs", CS("h2", 273, [H([1, 5], 0, 136, 0, 7), H([8, 12], 137, 136, 7, 7)]), "linear"),
    ("L1 2 hops x 2dmrs 200 PRB", CS("h2w", 273, [H([1, 5], 0, 200, 0, 7), H([8, 12], 73, 200, 7, 7)]), "linear"),
    ("L1 2 hops x 3dmrs 136 PRB", CS("h2d3", 273, [H([0, 3, 6], 0, 136, 0, 7), H([7, 10, 13], 137, 136, 7, 7)]), "linear"),
    ("L1 2 hops x 3dmrs 200 PRB", CS("h2d3w", 273, [H([0, 3, 6], 0, 200, 0, 7), H([7, 10, 13], 73, 200, 7, 7)]), "linear"),
    ("L1 2 hops x 4dmrs 150 PRB", CS("h2d4", 273, [H([0, 2, 4, 6], 0, 150, 0, 7), H([7, 9, 11, 13], 123, 150, 7, 7)]), "linear"),
    ("L1 2 hops x 2dmrs 40 PRB in 106", CS("h2n", 106, [H([1, 5], 0, 40, 0, 7), H([8, 12], 60, 40, 7, 7)]), "linear"),
    ("L1 2 hops x 1dmrs 12 PRB in 52", CS("h2t", 52, [H([2], 3, 12, 0, 7), H([9], 30, 12, 7, 7)]), "linear"),
    ("L1 2 hops x 2dmrs 80 PRB in 273", CS("h2m", 273, [H([1, 5], 0, 80, 0, 7), H([8, 12], 150, 80, 7, 7)]), "linear"),
    ("L2 2 hops x 2dmrs 136 PRB", CS("h2l2", 273, [H([1, 5], 0, 136, 0, 7), H([8, 12], 137, 136, 7, 7)], n_layers=2), "linear"),
    ("L4 2 hops x 2dmrs 136 PRB", CS("h2l4", 273, [H([1, 5], 0, 136, 0, 7, BOTH), H([8, 12], 137, 136, 7, 7, BOTH)], n_layers=4), "linear"),
    ("L2 2 hops x 2dmrs 12 PRB in 52", CS("h2l2n", 52, [H([1, 5], 3, 12, 0, 7), H([8, 12], 30, 12, 7, 7)], n_layers=2), "linear"),
    ("L1 type-2 mask filter", CS("t2", 273, [H([2, 11], 0, 273, re_masks=[S.TYPE2_CDM0])]), "linear"),
    ("L1 cnn in-painting", S.bench_case("filter", 1), "cnn"),
    ("L1 cnn type-2 mask", CS("t2c", 273, [H([2, 11], 0, 273, re_masks=[S.TYPE2_CDM0])]), "cnn"),
    ("L1 cnn type-2 100 PRB (iterated)", CS("t2c100", 273, [H([2, 11], 50, 100, re_masks=[S.TYPE2_CDM0])]), "cnn"),
    ("L2 25 PRB in 52", CS("small2", 52, [H([2, 11], 10, 25)], n_layers=2), "linear"),
    ("L4 25 PRB in 52", CS("small4", 52, [H([2, 11], 10, 25, re_masks=BOTH)], n_layers=4), "linear"),
    ("L2 66 PRB in 106", CS("mid2", 106, [H([2, 11], 20, 66)], n_layers=2), "linear"),
    ("L1 25 PRB in 52", CS("small", 52, [H([2, 11], 10, 25)]), "linear"),
    ("L1 66 PRB in 106", CS("mid", 106, [H([2, 11], 20, 66)]), "linear"),
    ("L1 6 PRB in 52", CS("tiny", 52, [H([2, 11], 10, 6)]), "linear"),
    ("L1 4dmrs 20 PRB in 52", CS("d4n", 52, [H([2, 5, 8, 11], 7, 20)]), "linear"),
    ("L1 4dmrs 150 PRB in 273", CS("d4w", 273, [H([2, 5, 8, 11], 60, 150)]), "linear"),
    ("L1 3dmrs 150 PRB in 273", CS("d3w", 273, [H([2, 7, 11], 60, 150)]), "linear"),
    ("L1 2dmrs 160 PRB in 273", CS("d2w", 273, [H([2, 11], 60, 160)]), "linear"),
    ("L1 3dmrs 70 PRB in 106", CS("d3m", 106, [H([2, 7, 11], 30, 70)]), "linear"),
    # the shapes the reference's own harness runs (52-PRB grids, ~3-PRB allocations, scripts/validation/validate_case{0,4,8}.py):
    # case 0 = 3 PRB at PRB 40, DM-RS symbols 0/4/8/12; case 4 = two hops of 3 PRB, BOTH described with the slot's whole symbol
    # range (validate_case4.py:85-103) -> hops whose fill rectangles share symbols; case 8 = two layers; + the same two-hop
    # convention at full band, the disjoint-symbol rows to compare with, and a 12-symbol grid (element-wise writer)
    ("harness case0: 3 PRB @40, 4dmrs, 52 grid", CS("hc0", 52, [H([0, 4, 8, 12], 40, 3)], scs=15e3), "linear"),
    ("harness case4: 2 hops x 3 PRB, full-slot hops", CS("hc4", 52, [H([0, 4], 3, 3, 0, 14), H([8, 12], 28, 3, 0, 14)], scs=15e3), "linear"),
    ("... same hops, disjoint symbols", CS("hc4d", 52, [H([0, 4], 3, 3, 0, 7), H([8, 12], 28, 3, 7, 7)], scs=15e3), "linear"),
    ("harness case8-like: L2 3 PRB 4dmrs", CS("hc8", 52, [H([0, 4, 8, 12], 40, 3)], n_layers=2, scs=15e3), "linear"),
    ("L2 2 hops x 3 PRB, full-slot hops", CS("hc4l2", 52, [H([0, 4], 3, 3, 0, 14), H([8, 12], 28, 3, 0, 14)], n_layers=2, scs=15e3), "linear"),
    ("L1 2 hops x 2dmrs 136 PRB, full-slot hops", CS("h2fs", 273, [H([1, 5], 0, 136, 0, 14), H([8, 12], 137, 136, 0, 14)]), "linear"),
    ("L1 2 hops x 1dmrs 12 PRB in 52, full-slot hops", CS("h2tfs", 52, [H([2], 3, 12, 0, 14), H([9], 30, 12, 0, 14)]), "linear"),
    ("L1 cnn 2 hops x 3 PRB, full-slot hops", CS("hc4c", 52, [H([0, 4], 3, 3, 0, 14), H([8, 12], 28, 3, 0, 14)], scs=15e3), "cnn"),
    ("L1 25 PRB in 52, 12-symbol grid", CS("s12", 52, [H([2], 10, 25, 0, 12)], n_sym=12), "linear"),
    ("L1 273 PRB, 12-symbol grid", CS("w12", 273, [H([2], 0, 273, 0, 12)], n_sym=12), "linear"),
]
