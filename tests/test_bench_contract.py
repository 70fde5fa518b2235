"""bench.py pieces that can be checked without a GPU: the roofline entry follows the contract
({bound, achieved, peak, unit, frac, traffic}) and picks the binding roof from the arithmetic intensity."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_entry_memory_bound_family():
    b = _bench()
    # 100 launches, 160 FLOP/B: below the bf16 ridge (2500e12 / 8000e9 = 312.5) -> HBM is the binding roof
    r = b.make_roofline('conv_igemm', flops=100 * 1.6e11, nbytes=100 * 1.0e9, total_ms=100 * 0.4, launches=100, steps=2,
                        mfma_peak_tflops=2500.0, traffic=1_100_000_000, traffic_src='profiles/x.json')
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['achieved'] - 2500.0) < 1e-6 and abs(r['frac'] - 2500.0 / 8000.0) < 1e-4
    assert r['traffic'] == 1_100_000_000 and r['launches_per_step'] == 50
    assert abs(r['achieved_tflops'] - 400.0) < 1e-6 and abs(r['mfma_frac'] - 0.16) < 1e-4
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r


def test_roofline_entry_compute_bound_family():
    b = _bench()
    r = b.make_roofline('conv_igemm', flops=10 * 4.0e11, nbytes=10 * 1.0e9, total_ms=10 * 0.5, launches=10, steps=1,
                        mfma_peak_tflops=2500.0, traffic=None, traffic_src=None)
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0
    assert abs(r['achieved'] - 800.0) < 1e-6 and abs(r['frac'] - 0.32) < 1e-4 and r['traffic'] is None
