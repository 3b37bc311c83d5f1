"""bench.py host logic that needs no GPU: the N > 1 self-launch (a CHILD torch.distributed.run,
started before torch or librau is imported -- never an exec from a process that touched the GPU)
and the strong-scaling argument handling."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_importing_bench_does_not_import_torch_or_load_librau():
    code = ("import sys, importlib.util as u; s = u.spec_from_file_location('b', 'bench.py'); "
            "m = u.module_from_spec(s); s.loader.exec_module(m); "
            "assert 'torch' not in sys.modules and 'rau_vqa_amd._lib' not in sys.modules")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_self_launch_starts_a_child_rendezvous_on_loopback(monkeypatch):
    b = load_bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(b.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--global-batch", "512"])
    with pytest.raises(SystemExit) as e:
        b.self_launch(types.SimpleNamespace(gpus=4))
    assert e.value.code == 7                                   # the child's code is relayed
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--global-batch", "512"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_rank_count_mismatch_and_indivisible_global_batch_are_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--global-batch", "7"], cwd=ROOT,
                       env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "not divisible" in r.stderr
