"""FROZEN table of the bf16-mode parity bounds of tests/test_model_gpu.py ("Next" 6b).

Written once, at the state of the end of round 5: (value measured on MI355X when the bound was set, bound).  ``_within`` in
test_model_gpu.py refuses a bound that differs from this table, so a test cannot be loosened by editing the number at its assert
alone: a change here needs a line in DESIGN.md section 2 with the per-tensor numbers before and after and the reason (rounds 1-5
moved two of them -- `tinyh.npz grad norms` 4.5e-2 -> 6e-2 and the config-3 click coincidence 3 -> 2 -- each time because another
GEMM instantiation shifted the bf16 operand roundings; both are recorded there).  Round 6 changed none of them.

Keys are matched as prefixes of the tag a test passes (tags may end in the name of the worst tensor).  The fp32 engine mode is not
in this table: it carries north_star's 1e-3 bound at every assert."""

UPPER = {
    # tag prefix: (measured, bound) -- relative errors
    "tiny.npz logits": (9.3e-3, 1.8e-2),
    "tiny.npz aux": (8.1e-3, 1.6e-2),
    "tiny.npz loss": (1.9e-4, 2e-3),
    "tiny.npz grad norms": (3.4e-2, 6e-2),
    "tinyh.npz logits": (1.41e-2, 2.6e-2),
    "tinyh.npz aux": (7.0e-3, 1.4e-2),
    "tinyh.npz loss": (5.6e-5, 2e-3),
    "tinyh.npz grad norms": (5.0e-2, 6e-2),
    "vitb box logits": (1.34e-2, 2.5e-2),
    "vitb box aux": (2.8e-3, 6e-3),
    "vitb loss": (1.8e-5, 1e-3),
    "vitb grad norms": (2.1e-2, 4e-2),
    "vitl8 box logits": (1.79e-2, 3.3e-2),
    "vitl8 loss": (2.1e-5, 1e-3),
    "vitl8 grad norms": (6.0e-2, 0.1),
    "bench-shape B=12 logits": (1.41e-2, 2.6e-2),
    "bench-shape B=4 logits": (1.41e-2, 2.6e-2),
    "bench-shape loss": (3.8e-4, 2e-3),
    "bench-shape B=12 bf16_rel as bench.py reports it": (1.555e-2, 2e-2),     # BENCH_r05.json parity.bf16_rel; bound asked by VERDICT r5
    "vitb click logits (budget test)": (1.25e-2, 2.3e-2),
}

LOWER = {
    # tag: (measured, floor) -- counts
    "config 3 bf16: leading clicks that coincide with the fp32 oracle's": (2, 2),
}


def frozen_upper(tag):
    best = None
    for k in UPPER:
        if tag.startswith(k) and (best is None or len(k) > len(best)):
            best = k
    return None if best is None else UPPER[best][1]
