"""bench.py: the figures its JSON line is built from (CPU only; the GPU run is the driver's). The algorithmic work per matrix and
per system is what DESIGN.md section 4 states, the defaults are the driver contract's (N = 1, a K/W that finishes in minutes),
and `--gpus N` without a launcher prepares one fresh process per rank before torch or HIP are loaded."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_work_matches_the_design_document():
    b = load_bench()
    n = 512
    assert b.getrf_flops(n) == 2.0 * n ** 3 / 3.0 - n ** 2 / 2.0 - n / 6.0
    flops, nbytes, launches = b.trailing_work(n)
    assert launches == 7
    assert abs(flops - 80.6e6) < 0.1e6 and abs(nbytes - 11.93e6) < 0.01e6          # 80.6 MFLOP, 11.9 MB
    ab = b.algorithmic_bytes(n, "linear_dense")
    assert ab["lu"] == 16 * n * n + 8 * n and ab["newton_iter"] == 8 * n * n + 40 * n + 8
    assert ab["sys"] == 16 * n * n + 40 * n and ab["sys_jac"] == 24 * n * n + 40 * n
    tim = {"lu": {"ms": 6.63e-3 * 1000, "systems": 1000, "launches": 1}, "newton_iter": {"ms": 0.549e-3 * 1000, "systems": 1000, "launches": 1}}
    lps = b.lu_plus_solve(tim, n, "unfused")
    assert abs(lps["getrf_us_per_matrix"] - 6.63) < 1e-9 and abs(lps["frac_of_hbm_peak"] - 0.1098) < 2e-4
    assert abs(lps["getrf_TFLOP/s"] - 13.47) < 0.02 and lps["valu_peak_TFLOP/s"] == 39.3
    assert "getrf_TFLOP/s" not in b.lu_plus_solve(tim, n, "unfused", dense=False)  # banded workload: bytes only


def test_help_runs_without_gpu_or_torch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in out.stdout


def test_self_launch_prepares_one_fresh_process_per_rank(monkeypatch):
    b = load_bench()
    started = []

    class FakeProc:
        def __init__(self, cmd, env):
            started.append((cmd, env))

        def wait(self):
            return 0

    import subprocess as sp
    monkeypatch.setattr(sp, "Popen", lambda cmd, env=None: FakeProc(cmd, env))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "5"])
    try:
        b.launch_ranks(3)
    except SystemExit as e:
        assert e.code == 0
    assert len(started) == 3
    ports = {env["MASTER_PORT"] for _, env in started}
    assert len(ports) == 1
    for r, (cmd, env) in enumerate(started):
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "3" and env["MASTER_ADDR"] == "127.0.0.1"
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "3", "--steps", "5"]


def test_eight_rank_start_rehearsed_on_the_host_side():
    """What `bench.py --gpus 8` needs from the HOST before its first barrier, at world size 8 (the driver's largest case) and a
    small N: eight fresh rank processes, each generating its own shard slice by slice as for the upload, handing its record to rank 0
    (through files: torch, which would open a GPU, is not imported); rank 0 reports seconds and peak resident memory per rank. No GPU is touched (`--inputs-only`), so this runs in the build
    container; the full-size figures measured on a GPU box's host are in DESIGN.md section 6."""
    import json
    import subprocess
    env = dict(os.environ, IDAHIP_GEN_PROCS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--inputs-only", "--n", "32", "--batch", "64"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["inputs_only"] and line["ranks"] == 8 and len(line["peak_rss_GiB_per_rank"]) == 8 and len(line["seconds_per_rank"]) == 8
    assert line["slices_per_rank"] >= 1 and line["seconds_until_every_rank_has_its_inputs_max"] >= 0.0
