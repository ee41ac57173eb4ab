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
    ("L1 25 PRB in 52", CS("small", 52, [H([2, 11], 10, 25)]), "linear"),
    ("L1 66 PRB in 106", CS("mid", 106, [H([2, 11], 20, 66)]), "linear"),
    ("L1 6 PRB in 52", CS("tiny", 52, [H([2, 11], 10, 6)]), "linear"),
    ("L1 4dmrs 20 PRB in 52", CS("d4n", 52, [H([2, 5, 8, 11], 7, 20)]), "linear"),
    ("L1 4dmrs 150 PRB in 273", CS("d4w", 273, [H([2, 5, 8, 11], 60, 150)]), "linear"),
    ("L1 3dmrs 150 PRB in 273", CS("d3w", 273, [H([2, 7, 11], 60, 150)]), "linear"),
    ("L1 2dmrs 160 PRB in 273", CS("d2w", 273, [H([2, 11], 60, 160)]), "linear"),
    ("L1 3dmrs 70 PRB in 106", CS("d3m", 106, [H([2, 7, 11], 30, 70)]), "linear"),
]
