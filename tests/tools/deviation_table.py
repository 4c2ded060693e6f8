"""Effect of each deliberate departure of the arithmetic contract from gsplat's written
formulas, measured on the committed float64-twin fixtures (CPU only; prints the table of DESIGN.md 3).

    python tests/tools/deviation_table.py            # markdown table of the fixtures
    python tests/tools/deviation_table.py --full     # + BASELINE configs 2, 3 and one view of 5 at FULL size: per departure
                                                     #   max |d rgb| and the NUMBER of pixels beyond 1e-4 / 1e-5 / 1e-6
Columns: max |d rgb| over all pixels of
  * the float32 oracle with all four textbook forms against the float64 twin  (float32 noise floor);
  * each contract form alone (the other three textbook) against the all-textbook float32 oracle;
  * the contract (all four forms) against the float64 twin.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle  # noqa: E402
from conftest import TWIN_CASES, load_twin_fixture, twin_scene_kwargs  # noqa: E402

ALL = oracle.VARIANT_TEXTBOOK_SIGMA | oracle.VARIANT_SIGMA_GUARD | oracle.VARIANT_T_PRODUCT | oracle.VARIANT_LIBM_EXP
FORMS = (("fused (dx, dy) sigma", oracle.VARIANT_TEXTBOOK_SIGMA), ("no sigma<0 guard", oracle.VARIANT_SIGMA_GUARD),
         ("T - alpha T", oracle.VARIANT_T_PRODUCT), ("polynomial exp", oracle.VARIANT_LIBM_EXP))
POLY = oracle.VARIANT_POLYNOMIAL_SIGMA   # study-only switch of the oracle: sigma as a polynomial in the tile-local pixel centre (the contract of rounds 1-4)
CORNER = 16   # ... on top of POLY: the polynomial about the tile's corner (the contract of rounds 1-3)


def render(g, mask):
    means, op, colors, kw = twin_scene_kwargs(g)
    W, H = [int(v) for v in g["wh"]]
    with oracle.variant(mask):
        return oracle.render(means, op, colors, g["viewmat"], g["K"], W, H, group_Rt=g["group_Rt"] if g["group_Rt"].size else None,
                             background=g["background"], **kw)


def measure(name):
    g = load_twin_fixture(name)
    text = render(g, ALL)
    row = {"fixture": name, "textbook f32 vs f64 twin": float(np.abs(text["rgb"] - g["rgb"]).max())}
    for label, bit in FORMS:
        one = render(g, ALL & ~bit)          # this form as in the contract, the others textbook
        row[label] = float(np.abs(one["rgb"] - text["rgb"]).max())
    contract = render(g, 0)
    row["contract vs f64 twin"] = float(np.abs(contract["rgb"] - g["rgb"]).max())
    row["contract vs textbook f32"] = float(np.abs(contract["rgb"] - text["rgb"]).max())
    return row


def count_row(label, img, ref):
    d = np.abs(img - ref).max(axis=2)
    return f"| {label} | {d.max():.1e} | {int((d > 1e-4).sum())} | {int((d > 1e-5).sum())} | {int((d > 1e-6).sum())} |"


def full_size(cfg, view=0):
    """One view of a BASELINE config at full size: every departure alone, and the contract, against the all-textbook
    float32 evaluation of the same lists (identical projection, binning and order: only T6's arithmetic differs)."""
    from sim_a_splat_amd.synthetic import NERFSTUDIO_EVAL_BACKGROUND as BG, config_scene_and_cameras
    sc, cams = config_scene_and_cameras(cfg)
    cam = cams[view]

    def r(mask):
        with oracle.variant(mask):
            return oracle.render_scene(sc, cam, background=BG)["rgb"]
    text = r(ALL)
    print(f"\nconfig {cfg}, view {view}: {sc.n} Gaussians, {cam.width}x{cam.height} = {cam.width * cam.height / 1e6:.2f} Mpixel; against the all-textbook float32 frame")
    print("| form | max d rgb | pixels > 1e-4 | > 1e-5 | > 1e-6 |\n|---|---|---|---|---|")
    for label, bit in FORMS:
        print(count_row(label + " alone", r(ALL & ~bit), text))
    print(count_row("**contract**", r(0), text))
    NS = ALL & ~oracle.VARIANT_TEXTBOOK_SIGMA
    print(count_row("polynomial sigma about the tile's centre alone (rounds 1-4's form)", r(NS | POLY), text))
    print(count_row("polynomial sigma about the tile's CORNER alone (rounds 1-3's form)", r(NS | POLY | CORNER), text))
    print(count_row("contract of round 4 (polynomial about the centre)", r(POLY), text))
    print(count_row("contract of rounds 1-3 (polynomial about the corner)", r(POLY | CORNER), text))


if __name__ == "__main__":
    if "--full" in sys.argv:
        for cfg in (2, 3, 5):
            full_size(cfg)
        sys.exit(0)
    rows = [measure(n) for n in TWIN_CASES]
    cols = list(rows[0].keys())
    print("| " + " | ".join(cols) + " |")
    print("|" + "---|" * len(cols))
    for r in rows:
        print("| " + " | ".join(r[c] if isinstance(r[c], str) else f"{r[c]:.1e}" for c in cols) + " |")
