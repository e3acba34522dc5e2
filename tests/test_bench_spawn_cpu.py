"""`bench.py --gpus N` started by hand must start its own ranks: the launch logic with the GPU count and the process
launcher replaced by stand-ins, and the real script on this GPU-less machine."""
import io
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_too_few_gpus_is_a_clear_error():
    bench = _bench()
    err = io.StringIO()
    calls = []
    rc = bench.spawn_ranks(2, ["--gpus", "2"], device_count=lambda: 1, run=lambda *a, **k: calls.append(a), err=err)
    assert rc != 0 and not calls
    assert "needs 2 GPUs, found 1" in err.getvalue()


def test_ranks_are_started_and_rank0_line_passed_through():
    bench = _bench()
    line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 4})
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout="noise from a rank\n" + line + "\n{\"not\": \"the result\"}\n", stderr="warn\n")

    out, err = io.StringIO(), io.StringIO()
    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "7"], device_count=lambda: 8, run=fake_run, out=out, err=err)
    assert rc == 0
    assert out.getvalue().strip() == line                       # exactly one line on stdout
    assert "noise from a rank" in err.getvalue() and "warn" in err.getvalue()
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert "RANK" not in seen["env"] or "RANK" in os.environ   # the children get their ranks from the launcher


def test_failed_ranks_and_missing_line_are_reported():
    bench = _bench()
    err = io.StringIO()
    rc = bench.spawn_ranks(2, [], device_count=lambda: 2,
                           run=lambda *a, **k: types.SimpleNamespace(returncode=3, stdout="", stderr="boom"), err=err)
    assert rc == 3 and "boom" in err.getvalue()
    err = io.StringIO()
    rc = bench.spawn_ranks(2, [], device_count=lambda: 2,
                           run=lambda *a, **k: types.SimpleNamespace(returncode=0, stdout="no json\n", stderr=""), err=err)
    assert rc != 0 and "without a result line" in err.getvalue()


def test_script_on_a_machine_without_enough_gpus():
    """The command form the driver uses for one GPU, asked for two, with no launcher and no RANK in the environment:
    a non-zero exit and the reason, not a rendezvous error."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "needs 2 GPUs, found" in r.stderr and r.stdout.strip() == ""
