"""bench.py's CPU leg (the only part of the bench that may touch oracle/): runs on the tiny geometry and
returns the fields the bench line's `cpu_baseline` object carries."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_cpu_baseline_fields():
    import bench
    from vita_amd.config import VitaConfig
    r = bench.cpu_baseline(VitaConfig.tiny(), n_layers=2, ctx=16, n_tok=3)
    assert r["kind"] == "port" and r["unit"] == "tokens/s" and r["value"] > 0
    assert str(r["cores"]) in r["tokens_per_s_by_threads"]
    assert r["value"] == max(r["tokens_per_s_by_threads"].values())


def test_pmc_traffic_reads_committed_profile():
    import bench
    v = bench.pmc_traffic("k_dec_gateup")
    assert v is not None and 0.9 < v / 469827584 < 1.2     # HBM bytes per launch ~ the algorithmic bytes
    assert bench.pmc_traffic("no_such_kernel") is None
