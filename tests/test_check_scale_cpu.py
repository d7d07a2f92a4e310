"""tools/check_scale.py: the self-check of a multi-GPU bench line against DESIGN.md 6's predictions."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_scale", os.path.join(ROOT, "tools", "check_scale.py"))
cs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cs)


def _line(n, ms, comm=None):
    return {"metric": "x", "n_gpus": n, "scaling": "strong", "ms_per_step": ms, "value": 240.0 / (ms * 1e-3),
            "config": {"global_batch": 240}, "comm": comm}


def _comm(n, **kw):
    c = {"backend": "nccl", "rccl_ranks": n, "collectives_per_step": 14, "exposed_exchange_ms": 0.8,
         "step_ms_without_exchange": cs.EXPECT[n][0], "busbw_frac_of_xgmi": 0.3}
    c.update(kw)
    return c


def test_lines_inside_the_predictions_pass():
    assert cs.check(_line(1, 43.5)) == []
    assert cs.check(_line(8, 10.0, _comm(8))) == []
    assert cs.check(_line(4, 15.4, _comm(4, exposed_exchange_ms=1.6))) == []


def test_deviations_are_reported():
    assert any("RCCL saw 1 ranks" in m for m in cs.check(_line(8, 10.9, _comm(8, rccl_ranks=1))))
    assert any("not 'nccl'" in m for m in cs.check(_line(8, 10.9, _comm(8, backend="gloo"))))
    assert any("collectives per step" in m for m in cs.check(_line(8, 10.9, _comm(8, collectives_per_step=27))))
    assert any("exposed exchange" in m for m in cs.check(_line(8, 10.9, _comm(8, exposed_exchange_ms=5.0))))
    assert any("ms per step" in m for m in cs.check(_line(8, 20.0, _comm(8))))
    assert cs.EXPECT[8][1] == 14 and cs.EXPECT[2][1] == 14                 # from unimm_amd.bucket_plan
    assert any("no `comm` block" in m for m in cs.check(_line(2, 24.0)))
    bad = _line(8, 10.9, _comm(8))
    bad["value"] = 99999.0
    assert any("is not global_batch / ms_per_step" in m for m in cs.check(bad))
