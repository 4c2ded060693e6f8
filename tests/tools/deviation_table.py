"""Effect of each deliberate departure of the arithmetic contract from gsplat's written
formulas, measured on the committed float64-twin fixtures (CPU only; prints the table of DESIGN.md 3).

    python tests/tools/deviation_table.py            # markdown table
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
FORMS = (("polynomial sigma", oracle.VARIANT_TEXTBOOK_SIGMA), ("no sigma<0 guard", oracle.VARIANT_SIGMA_GUARD),
         ("T - alpha T", oracle.VARIANT_T_PRODUCT), ("polynomial exp", oracle.VARIANT_LIBM_EXP))


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


if __name__ == "__main__":
    rows = [measure(n) for n in TWIN_CASES]
    cols = list(rows[0].keys())
    print("| " + " | ".join(cols) + " |")
    print("|" + "---|" * len(cols))
    for r in rows:
        print("| " + " | ".join(r[c] if isinstance(r[c], str) else f"{r[c]:.1e}" for c in cols) + " |")
