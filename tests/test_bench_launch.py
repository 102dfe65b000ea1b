"""`python bench.py --gpus N` starts its N ranks itself (the form the round driver uses; there is no torchrun around it).

CPU part: the launcher's own contract -- argument handling, the environment it gives the ranks, a dying rank stops the others
and its exit code comes back.  GPU part: the exact command of the review item on the one-GPU box (ranks share cuda:0 and exchange
over gloo; RCCL needs one device per rank and runs on the driver's multi-GPU node)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_gpu_is_a_loud_exit_not_a_hint():
    """Without a GPU the command exits non-zero and says why; it no longer answers with a torchrun recipe."""
    if torch.cuda.is_available():
        pytest.skip("GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "no GPU visible" in r.stderr and "torch.distributed.run" not in r.stderr
    assert r.stdout.strip() == ""


def test_launcher_environment_and_failure_propagation(monkeypatch, tmp_path):
    """self_launch(N): N children of THIS command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT;
    fewer devices than ranks -> gloo + the shared-GPU marker; the first non-zero exit stops the rest and is returned."""
    bench = _load_bench()
    log = tmp_path / "ranks.jsonl"
    child = tmp_path / "child.py"
    # a stand-in for the rank processes: records its environment; rank 1 fails, rank 0 would otherwise sleep for a minute
    child.write_text(
        "import json, os, sys, time\n"
        "keys = ['RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'LVAE_DIST_BACKEND', 'LVAE_SHARED_GPU', 'LVAE_BENCH_LAUNCHER']\n"
        "open(%r, 'a').write(json.dumps({k: os.environ.get(k) for k in keys} | {'argv': sys.argv[1:]}) + '\\n')\n"
        "if os.environ['RANK'] == '1':\n"
        "    time.sleep(0.5); sys.exit(7)\n"
        "time.sleep(60)\n" % str(log))
    monkeypatch.setattr(bench, "__file__", str(child))
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "5"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LVAE_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    import time
    t0 = time.time()
    rc = bench.self_launch(3, deadline_s=60, attempts=3)      # fewer devices than ranks: gloo functional check, ONE attempt
    assert rc == 7
    assert time.time() - t0 < 30            # ranks 0 and 2 were terminated, not waited for
    recs = [json.loads(l) for l in log.read_text().splitlines()]
    assert sorted(r["RANK"] for r in recs) == ["0", "1", "2"]
    for r in recs:
        assert r["LOCAL_RANK"] == r["RANK"] and r["WORLD_SIZE"] == "3" and r["MASTER_ADDR"] == "127.0.0.1"
        assert int(r["MASTER_PORT"]) > 0 and r["argv"] == ["--gpus", "3", "--steps", "5"]
        assert r["LVAE_DIST_BACKEND"] == "gloo" and "3 ranks on 1 GPU" in r["LVAE_SHARED_GPU"] and r["LVAE_BENCH_LAUNCHER"] == "self"
    assert len({r["MASTER_PORT"] for r in recs}) == 1


def test_launcher_uses_rccl_when_every_rank_has_a_device(monkeypatch, tmp_path):
    bench = _load_bench()
    log = tmp_path / "ranks.jsonl"
    child = tmp_path / "child.py"
    child.write_text("import json, os\nopen(%r, 'a').write(json.dumps({k: os.environ.get(k) for k in ['RANK', 'LVAE_DIST_BACKEND', 'LVAE_SHARED_GPU', "
                     "'HSA_ENABLE_IPC_MODE_LEGACY']}) + '\\n')\n" % str(log))
    monkeypatch.setattr(bench, "__file__", str(child))
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("LVAE_DIST_BACKEND", raising=False)
    assert bench.self_launch(4) == 0
    recs = [json.loads(l) for l in log.read_text().splitlines()]
    assert len(recs) == 4
    for r in recs:      # nothing overrides the default backend (nccl = RCCL wherever CUDA is available), dmabuf IPC is on
        assert r["LVAE_DIST_BACKEND"] is None and r["LVAE_SHARED_GPU"] is None and r["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _stand_in(bench, monkeypatch, tmp_path, body, ndev=8):
    child = tmp_path / "child.py"
    child.write_text(body)
    monkeypatch.setattr(bench, "__file__", str(child))
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: ndev)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LVAE_DIST_BACKEND", "LVAE_BENCH_WORKER", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)


def test_hung_ranks_become_a_json_error_line_not_silence(monkeypatch, tmp_path, capfd):
    """ranks that never finish are stopped at the deadline, the next rung of the schedule ladder is tried, and
    when every attempt has failed the SUPERVISOR prints one JSON line {"error", "n_gpus", "attempts", "ranks_alive"} with rc != 0."""
    bench = _load_bench()
    log = tmp_path / "ranks.jsonl"
    _stand_in(bench, monkeypatch, tmp_path,
              "import json, os, time\n"
              "open(%r, 'a').write(json.dumps({k: os.environ.get(k) for k in ['RANK', 'LVAE_BENCH_ATTEMPT', 'LVAE_BENCH_SCHEDULE', "
              "'LVAE_DP_CONSERVATIVE', 'LVAE_BENCH_PERSISTENT', 'MASTER_PORT', 'LVAE_BENCH_WORKER', 'LVAE_DIST_TIMEOUT']}) + '\\n')\n"
              "time.sleep(600)\n" % str(log))
    import time
    t0 = time.time()
    rc = bench.self_launch(2, deadline_s=2.0, attempts=3)
    assert rc == 124 and time.time() - t0 < 40
    out = capfd.readouterr()
    lines = [l for l in out.out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    err = json.loads(lines[0])
    assert err["error"] and err["n_gpus"] == 2 and err["value"] is None and len(err["attempts"]) == 3
    assert err["ranks_alive"] == [0, 1] and all("timeout" in a["outcome"] for a in err["attempts"])
    assert [a["schedule"] for a in err["attempts"]] == [r[0] for r in bench.LAUNCH_LADDER]
    recs = [json.loads(l) for l in log.read_text().splitlines()]
    assert len(recs) == 6
    by_attempt = {a: [r for r in recs if r["LVAE_BENCH_ATTEMPT"] == str(a)] for a in range(3)}
    assert by_attempt[0][0]["LVAE_DP_CONSERVATIVE"] is None and by_attempt[0][0]["LVAE_BENCH_PERSISTENT"] is None
    assert by_attempt[1][0]["LVAE_DP_CONSERVATIVE"] == "1" and by_attempt[1][0]["LVAE_BENCH_PERSISTENT"] is None
    assert by_attempt[2][0]["LVAE_DP_CONSERVATIVE"] == "1" and by_attempt[2][0]["LVAE_BENCH_PERSISTENT"] == "0"
    for a in range(3):      # both ranks of an attempt share its rendezvous port
        assert len({r["MASTER_PORT"] for r in by_attempt[a]}) == 1 and all(r["LVAE_BENCH_WORKER"] == "1" for r in by_attempt[a])
    assert "attempt 2 of 3" in out.err and "conservative-exchange" in out.err


def test_second_rung_of_the_ladder_rescues_the_run(monkeypatch, tmp_path, capfd):
    """The default schedule dies (rank 1 exits 9), the conservative one works: rc 0, rank 0's line is the only JSON line and says
    which attempt it came from."""
    bench = _load_bench()
    _stand_in(bench, monkeypatch, tmp_path,
              "import json, os, sys, time\n"
              "if os.environ.get('LVAE_DP_CONSERVATIVE') != '1':\n"
              "    if os.environ['RANK'] == '1': sys.exit(9)\n"
              "    time.sleep(600)\n"
              "if os.environ['RANK'] == '0':\n"
              "    print(json.dumps({'value': 1.0, 'attempt': os.environ['LVAE_BENCH_ATTEMPT'], 'prior': json.loads(os.environ['LVAE_BENCH_PRIOR'])}), flush=True)\n"
              "    open(os.environ['LVAE_BENCH_DONE'], 'w').write('done')\n")
    assert bench.self_launch(2, deadline_s=30.0, attempts=3) == 0
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["attempt"] == "1" and rec["prior"][0]["schedule"] == "default" and "rank 1 exited with 9" in rec["prior"][0]["outcome"]


def test_a_hang_after_the_result_line_is_not_a_failure(monkeypatch, tmp_path, capfd):
    """Rank 0 prints its line, writes the marker, then teardown hangs: the supervisor stops the ranks at the deadline, returns 0
    and starts no second attempt (one JSON line)."""
    bench = _load_bench()
    _stand_in(bench, monkeypatch, tmp_path,
              "import json, os, time\n"
              "if os.environ['RANK'] == '0':\n"
              "    print(json.dumps({'value': 2.0}), flush=True)\n"
              "    open(os.environ['LVAE_BENCH_DONE'], 'w').write('done')\n"
              "time.sleep(600)\n")
    assert bench.self_launch(2, deadline_s=3.0, attempts=3) == 0
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert lines == [json.dumps({"value": 2.0})]


def test_under_torchrun_every_launcher_child_supervises_its_own_rank(monkeypatch, tmp_path, capfd):
    """torch.distributed.run form: WORLD_SIZE / RANK come from the launcher; the process becomes the supervisor of its own rank.
    Attempt 0 keeps the launcher's rendezvous untouched; attempt 1 moves to MASTER_PORT + 1 with rank 0 hosting the store."""
    bench = _load_bench()
    log = tmp_path / "ranks.jsonl"
    _stand_in(bench, monkeypatch, tmp_path,
              "import json, os, sys\n"
              "open(%r, 'a').write(json.dumps({k: os.environ.get(k) for k in ['RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', "
              "'TORCHELASTIC_USE_AGENT_STORE', 'LVAE_BENCH_ATTEMPT', 'LVAE_BENCH_LAUNCHER']}) + '\\n')\n"
              "sys.exit(0 if os.environ['LVAE_BENCH_ATTEMPT'] == '1' else 5)\n" % str(log))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("MASTER_PORT", "29871")
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    assert bench.supervise_own_rank(2, deadline_s=2.0, attempts=3) == 0
    recs = [json.loads(l) for l in log.read_text().splitlines()]
    assert [r["LVAE_BENCH_ATTEMPT"] for r in recs] == ["0", "1"] and all(r["RANK"] == "1" and r["WORLD_SIZE"] == "2" for r in recs)
    assert recs[0]["MASTER_PORT"] == "29871" and recs[0]["TORCHELASTIC_USE_AGENT_STORE"] == "True"
    assert recs[1]["MASTER_PORT"] == "29872" and recs[1]["TORCHELASTIC_USE_AGENT_STORE"] == "False"
    assert all(r["LVAE_BENCH_LAUNCHER"] == "torch.distributed.run" for r in recs)
    assert not [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]      # rank 1's supervisor never prints a line


@pytest.mark.parametrize("n", [4])
def test_bench_gpus_4_completes_on_the_emulator_over_gloo(n):
    """`bench.py --gpus 4` on one shared device did not finish in 10 minutes while 2 and 8 did.  The same
    command on the CPU emulator (LVAE_BENCH_EMU=1: the test hook that runs this file's multi-rank control flow on device "cpu" over
    gloo) completes in seconds with identical replicas -- the launcher, the exchange and the loop bookkeeping are not what hangs
    at world 4 (tests/test_dist_gloo.py has the gradient identity at world 4 and 8)."""
    if torch.cuda.is_available():
        pytest.skip("CPU emulator leg")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LVAE_BENCH_EMU"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "3", "--warmup", "1", "--workload", "toy", "--dtype", "f32", "--pool", "4"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["config"]["global_batch"] == 16 * n and "test hook" in out["config"]["dp_transport"]
    assert out["dp_breakdown"]["replicas_identical"] is True and len(out["dp_breakdown"]["per_rank"]) == n
    # (normally the first attempt, 15 s; on a CI box busy with something else an attempt may run into its deadline and the
    #  supervisor's next rung completes the run -- which is what the supervisor is for)
    assert out["launch"]["attempt"] >= 1 and out["launch"]["schedule"]


def test_the_drivers_torchrun_command_end_to_end_on_the_emulator():
    """The round driver's multi-GPU form, literally: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- here on the CPU emulator over gloo (LVAE_BENCH_EMU test hook).
    Every launcher child becomes the supervisor of its own rank (attempt 0 on the launcher's own rendezvous / agent store); rank 0's
    worker prints the one line."""
    if torch.cuda.is_available():
        pytest.skip("CPU emulator leg")
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(LVAE_BENCH_EMU="1", LVAE_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "toy", "--dtype", "f32",
                        "--pool", "4"], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["config"]["launcher"] == "torch.distributed.run"
    assert out["launch"]["attempt"] == 1 and out["dp_breakdown"]["replicas_identical"] is True


@pytest.mark.gpu
def test_bench_gpus_2_runs_end_to_end_on_this_box():
    """VERDICT r4 item 1, literally: `python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline` -> rc 0, one JSON line with
    n_gpus 2, dp_breakdown and every rank's ladder rung."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["value"] > 0
    assert out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp2"
    bd = out["dp_breakdown"]
    assert len(bd["per_rank"]) == 2 and len(bd["lstm_ladder_rung_per_rank"]) == 2
    if torch.cuda.device_count() < 2:
        assert bd["backend"] == "gloo" and "functional check" in out["config"]["dp_transport"]
        assert "NOT a scaling measurement" in r.stderr
    else:
        assert bd["backend"] == "nccl" and "RCCL" in out["config"]["dp_transport"]
    assert bd["replicas_identical"] is True


@pytest.mark.gpu
def test_rccl_run_keeps_rung_0_and_identical_replicas_on_a_multi_gpu_box():
    """on a box with >= 2 GPUs the ranks exchange over RCCL; after 50 steps every rank still runs the persistent
    launches with the XCD-local hand-off (rung 0: no collective took compute units from a persistent launch), the replicas hold the
    same bits, and the line came from the FIRST rung of the launch ladder."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL: one communicator rank per device)")
    n = min(8, torch.cuda.device_count())
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "50", "--warmup", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    bd = out["dp_breakdown"]
    assert out["n_gpus"] == n and bd["backend"] == "nccl"
    assert bd["lstm_ladder_rung_per_rank"] == [0] * n, bd["lstm_ladder_rung_per_rank"]
    assert bd["replicas_identical"] is True
    assert out["launch"]["attempt"] == 1 and out["launch"]["schedule"] == "default", out["launch"]
