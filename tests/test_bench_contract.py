"""CPU-side guards of bench.py's contract pieces that do not need a GPU: the committed PMC traffic resolves for the
dominant kernels of the default configuration (round 2 shipped `roofline.traffic: null` because the newest
r*_traffic.json was the Modular kernels' file), and the per-kernel byte table names what the library times."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_traffic_resolves_for_the_default_config():
    b = _bench()
    for dominant in ("k1_vardct", "k23_fused_filters"):
        traffic, path = b.resolve_traffic(dominant)
        assert traffic is not None and path is not None, dominant
        assert os.path.exists(os.path.join(ROOT, path))
        # an 8192^2 frame: at least the algorithmic bytes, and not absurdly more
        algo = b.ALGO_BYTES_PER_PX[dominant] * 8192 * 8192
        assert 0.9 * algo < traffic < 3 * algo, (dominant, traffic, algo)
    assert b.resolve_traffic("no_such_kernel") == (None, None)


def test_kernel_byte_table_matches_the_library_timer_names():
    b = _bench()
    import glob
    srcs = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "jxl_rs_amd", "csrc", "abi_*.hip")))
    for name in b.ALGO_BYTES_PER_PX:
        assert f'"{name}"' in srcs or name.startswith("k3"), name  # k3a_/k3b_/k3c_ names are composed at run time


def test_chain_traffic_resolves_for_both_resident_forms():
    """top-level `chain_counter_traffic_bytes` (dense slabs resident) and the slot-resident leg's counter traffic come
    from committed counter files of their own workloads; the slot-bucketed form must move less than the dense one"""
    b = _bench()
    dense, f_dense = b.chain_traffic("8192 d1")
    slots, f_slots = b.chain_traffic("8192 d1 slots")
    assert dense and slots and f_dense != f_slots
    ideal = b.FUSED_IDEAL_BYTES_PER_PX * 8192 * 8192
    assert ideal < slots < dense < 3 * ideal, (ideal, slots, dense)
    assert b.chain_traffic("no such workload") == (None, None)


def test_gpus_n_without_devices_refuses_instead_of_measuring_one_rank():
    """`python bench.py --gpus 2` on a box without two devices must die naming what is missing, not
    print an n_gpus: 1 line"""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: no device visible to this child
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "device" in r.stderr, r.stderr[-500:]
    assert '"n_gpus"' not in r.stdout
