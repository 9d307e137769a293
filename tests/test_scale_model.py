"""Host logic (no GPU): the analytic step model of the sharded factorisation (tools/scale_model.py) — the PREDICTION the docs quote for the
first multi-GPU run (LABBOOK section 0'', README) must be what the committed tool prints from its recorded inputs, and the model must
behave like a model of strong scaling: efficiency 1 on one GPU, never above 1, falling with the GPU count, and better at the larger size."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_prediction_is_reproducible_and_sane():
    newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_scale_model.json"))[-1]   # r06_scale_model.json
    committed = json.load(open(os.path.join(ROOT, "profiles", newest)))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "m.json")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "scale_model.py"), "--chain-ms", str(committed["inputs"]["chain_ms_per_1024_block"]), "--out", out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        again = json.load(open(out))
    assert again == committed
    rows = {k: v for k, v in committed.items() if isinstance(v, list)}
    assert len(rows) == 2
    eff = {}
    for name, table in rows.items():
        assert [r["gpus"] for r in table] == [1, 2, 4, 8]
        e = [r["strong_scaling_efficiency"] for r in table]
        assert e[0] == 1.0 and all(0.0 < x <= 1.0 for x in e) and all(a >= b for a, b in zip(e, e[1:])), (name, e)
        fits = [r["fits_per_sec"] for r in table]
        assert all(b > a for a, b in zip(fits, fits[1:])), (name, fits)       # more GPUs are still faster, only less than linearly
        for r in table:
            assert abs(sum(r["parts_s"].values()) - r["fit_s"]) <= 0.02 * r["fit_s"], (name, r["gpus"])   # the parts account for the fit
        eff[name] = e[-1]
    small = next(v for k, v in eff.items() if "50000" in k)
    large = next(v for k, v in eff.items() if "200000" in k)
    assert large > small          # the update grows as N^3, the chain and the exchanges as N^2: the larger problem scales better
