"""Regenerates tests/golden/svgf_golden.npz from the CPU oracle (oracle/*.c).

SVGF parity is UNPINNED BY THE REFERENCE (it has no SVGF code or vectors): these
vectors pin the oracle against itself so an accidental change of the spec implementation shows
up, and give the GPU tests a second, committed target.  Inputs are the deterministic synthetic
scene (seed 1234, 64x48, frames 0..2) and a 96x80 Cornell crop, so only outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as orc  # noqa: E402

W, H, FRAMES = 64, 48, 3
CROP = (slice(200, 280), slice(150, 246))  # rows, cols of the Cornell planes (80 x 96)


def run_sequence(width, height, inputs, params):
    """inputs: list of (color, nd, motion).  Returns dict of per-frame outputs."""
    out = {}
    hist_c = hist_m = prev_nd = None
    for i, (color, nd, motion) in enumerate(inputs):
        fr = orc.Frame(width, height, color, nd, motion, hist_c, hist_m, prev_nd)
        orc.frame(fr, params)
        out[f"f{i}_t_color"] = fr.t_color.copy()
        out[f"f{i}_t_moments"] = fr.t_moments.copy()
        out[f"f{i}_t_len"] = fr.t_len.copy()
        out[f"f{i}_t_debug"] = fr.t_debug.copy()
        out[f"f{i}_v_color"] = fr.v_color.copy()
        out[f"f{i}_hist_color_out"] = fr.hist_color_out.copy()
        out[f"f{i}_out_color"] = fr.out_color.copy()
        hist_c, hist_m, prev_nd = fr.history()
    return out


def main():
    p = orc.default_params()
    synth = [orc.synth_gbuffer(W, H, f) for f in range(FRAMES)]
    g = {"synth_" + k: v for k, v in run_sequence(W, H, synth, p).items()}
    g["synth_color0_sum"] = np.array([synth[0][0].astype(np.float64).sum()])
    color, nd, motion = orc.cornell_svgf_inputs()
    color, nd, motion = color[CROP].copy(), nd[CROP].copy(), motion[CROP].copy()
    ch, cw = color.shape[:2]
    g.update({"cornell_" + k: v for k, v in run_sequence(cw, ch, [(color, nd, motion)] * 2, p).items()})
    path = os.path.join(HERE, "svgf_golden.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
