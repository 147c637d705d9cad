"""bench.py --gpus N must start N ranks by itself (the flag was parsed and ignored).  CPU tier: the wiring only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_builds_a_one_rank_per_gpu_launch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--print-launch"],
                         capture_output=True, text=True, timeout=120, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert out.returncode == 0, out.stderr
    cmd = out.stdout.split()
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1"]          # the ranks see the same flags


def test_launch_command_is_importable_and_uses_a_free_port():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.launch_command(2, ["--gpus", "2"])
    b = bench.launch_command(2, ["--gpus", "2"])
    assert a[a.index("--master-port") + 1].isdigit() and b[b.index("--master-port") + 1].isdigit()
    assert "--nproc-per-node=2" in a


def test_mismatched_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
