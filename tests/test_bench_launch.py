"""bench.py started plainly with --gpus N > 1 (no torch.distributed.run in front): the parent spawns the N ranks itself.
What is checked here, on the CPU: the child command lines and their environment; that the parent relays rank 0's stdout, waits for every
rank and reports the first failing rank's exit code; that it starts children before anything imports torch (a process that has touched the
GPU must not be replaced — it is not: the ranks are child processes, the parent never execs)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_child_command_lines():
    import bench
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2", "--config", "cfg3"]
    env = {"PATH": "/usr/bin", "MFVI_BENCH_BACKEND": "gloo"}
    kids = bench.child_commands(4, argv, env, port=29517)
    assert len(kids) == 4
    for r, (cmd, e) in enumerate(kids):
        assert cmd[0] == sys.executable and os.path.samefile(cmd[1], os.path.join(ROOT, "bench.py")) and cmd[2:] == argv
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == "4"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29517"
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # dmabuf IPC: RCCL across processes needs it on this pool
        assert e["MFVI_BENCH_BACKEND"] == "gloo" and e["PATH"] == "/usr/bin"      # the caller's environment travels
    assert "RANK" not in env                                   # the parent's own environment is left alone
    # without a port the parent picks a free one, the same for every rank
    kids = bench.child_commands(2, argv, env)
    assert kids[0][1]["MASTER_PORT"] == kids[1][1]["MASTER_PORT"] and int(kids[0][1]["MASTER_PORT"]) > 0


def _fake_bench(tmp_path, body):
    """A stand-in for bench.py that keeps self_launch / child_commands and replaces what a rank does."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    p = tmp_path / "bench.py"
    p.write_text(head + textwrap.dedent(body))
    return str(p)


RANK_BODY = """
def main():
    import argparse
    ap = argparse.ArgumentParser(); ap.add_argument("--gpus", type=int, default=1); ap.add_argument("--fail-rank", type=int, default=-1)
    args = ap.parse_args()
    assert "torch" not in sys.modules                      # nothing touched torch (or the GPU) before the decision to spawn
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)       # the ranks really meet: MASTER_ADDR / MASTER_PORT are consistent
    import torch
    t = torch.tensor([float(rank + 1)]); dist.all_reduce(t)
    if rank == args.fail_rank:
        os._exit(3)
    print("noise from rank %d" % rank) if rank else print(json.dumps({"rccl_ranks": dist.get_world_size(), "sum": float(t)}))
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
"""


def test_parent_spawns_ranks_and_relays_rank0(tmp_path):
    fake = _fake_bench(tmp_path, RANK_BODY)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, fake, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]      # (gloo's own connection banner)
    assert len(lines) == 1, r.stdout                        # ONE JSON line: rank 0's; the other ranks' stdout is dropped
    assert json.loads(lines[0]) == {"rccl_ranks": 2, "sum": 3.0}


def test_parent_reports_a_failing_rank(tmp_path):
    fake = _fake_bench(tmp_path, RANK_BODY)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, fake, "--gpus", "2", "--fail-rank", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3
    assert "rank 1 exited with code 3" in r.stderr


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)
