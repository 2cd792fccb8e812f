"""bench.py's driver contract, end to end on the GPU box: the N = 1 line, `bench.py --gpus 2` launched DIRECTLY (it spawns its
own ranks) and the same through torch.distributed.run -- both ranks on this GPU over the library's socket transport (a one-GPU
box cannot host two RCCL ranks; the RCCL calls are exercised single-rank through the C ABI, tools/gpu_nccl_selftest.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline")


def _last_json(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def im_peak_ok(im):
    pk = im["peak_same_run"]
    return 10 < pk["peak_T"] < 80 and im["peak"] == pk["peak_T"] == max(pk["runs_T"])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--logn", "16", "--cpu-sample-logn", "10",
                        "--verify-steps", "3", "--batch", "256"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in KEYS + ("cpu_baseline",):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["value"] > 0
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["result_equals_closed_form"] is True
    assert im_peak_ok(d["roofline_int_mad"])
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] == 1
    assert "workload" in d["config"]
    im = d["roofline_int_mad"]
    assert im["mixed_adds_per_launch"] == im["bucket_entries"] - im["chunks"] > 0 and 0 < im["frac"] < 1
    assert 0.3 < im["madd_probe"]["k_accumulate_vs_probe"] < 1.5      # sanity only: both sides are short timings
    assert d["config"]["device_build"] == "staged"
    assert set(d["seeds_ms_per_step"]) >= {"1", "2", "3", "min", "median"}
    assert [b["cores"] for b in d["cpu_baseline"]["stronger_non_reference_baselines"]][0] == 1
    # the metric's second half, in the same line
    sec = d["secondary"]
    assert sec["unit"] == "proofs/s" and sec["value"] > 0 and sec["batch"] == 256 and sec["distinct_proofs"] >= 64
    assert 0 < sec["roofline_int_mad"]["frac"] < 1 and sec["roofline_int_mad"]["kernel"].startswith("k_batch_decompress")
    assert sec["cpu_baseline"]["kind"] == "port" and sec["cpu_baseline"]["cores"] == 1


def _check_two_rank_line(d):
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["terms_total"] == 2 * (1 << 16)
    assert d["config"]["parallelism"].startswith("windows x2 . points x1")
    assert d["config"]["result_equals_closed_form"] is True
    col = d["collective"]
    assert col["world_seen"] == 2 and col["backend"] == "socket" and col["bytes_per_step"] == 288 and col["ms_per_exchange"] > 0
    assert d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"]
    assert d["strong_2_22_total"]["terms_total"] == 1 << 22 and d["strong_2_22_total"]["value"] > 0
    assert d["windows_only"]["same_result_as_default_shard"] is True if "windows_only" in d else True


def test_two_rank_direct_launch_prints_one_line():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts its own two ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--logn", "16", "--same-device", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    _check_two_rank_line(_last_json(r.stdout))


def test_two_rank_torchrun_launch_prints_one_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--logn", "16", "--same-device", "--no-cpu-baseline", "--no-secondary"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["collective"]["world_seen"] == 2 and d["config"]["result_equals_closed_form"] is True


def test_two_rank_verify_mode_reports_per_rank_front_end():
    """config 5's structure at N = 2 on this GPU: per-rank front-end times and thread counts are in the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "verify", "--steps", "3", "--warmup", "1",
                        "--batch", "256", "--same-device", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["unit"] == "proofs/s" and d["value"] > 0
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and len(pr["front_end_ms_per_step"]) == 2 and pr["host_threads"] >= 1


def test_rccl_through_the_c_abi_single_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_nccl_selftest.py")], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0 and "rccl selftest ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_rccl_refusal_falls_back_to_the_socket_transport():
    """Two ranks on ONE device with the RCCL backend forced: ncclCommInitRank refuses (duplicate GPU).  Every rank must notice,
    start over on the socket transport and say so in the line -- not hang, not die."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["CG1_BENCH_TRY_RCCL_ON_ONE_DEVICE"] = "1"
    env["NCCL_DEBUG"] = "WARN"
    r = subprocess.run(["timeout", "-k", "10", "400", sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--logn", "14",
                        "--same-device", "--backend", "rccl", "--no-cpu-baseline", "--no-secondary"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    col = d["collective"]
    assert col["backend"] == "socket" and "RCCL not attached" in col["backend_note"] and col["world_seen"] == 2
    assert d["config"]["result_equals_closed_form"] is True


def test_four_rank_rehearsal_of_both_modes():
    """The N > 1 path with more than two ranks: 4 ranks on this one GPU (the pool allows at most 6 processes on a card, this process
    is one of them), hybrid sharding = 2 window groups x 2 point groups, then the proof-per-GPU verify mode.  Rehearsal, not a
    scaling measurement: the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run(["timeout", "-k", "10", "500", sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--logn", "14", "--same-device", "--no-cpu-baseline", "--no-secondary"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 4 and d["collective"]["world_seen"] == 4 and d["config"]["result_equals_closed_form"] is True
    assert d["config"]["parallelism"].startswith("windows x2 . points x2")
    assert "REHEARSING" in d["measured_on"] and d["collective"]["rendezvous_s"] >= 0
    r = subprocess.run(["timeout", "-k", "10", "500", sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--mode", "verify", "--steps", "2", "--warmup", "1",
                        "--batch", "128", "--same-device", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 4 and d["unit"] == "proofs/s" and len(d["per_rank"]["ms_per_step"]) == 4
