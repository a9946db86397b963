"""`python bench.py --gpus N` starts its own N ranks (reference README.md:67 starts 8 processes with one command): the
parent builds a torch.distributed.run command line BEFORE anything touches the GPU, runs it as a child process, relays
rank 0's JSON line on stdout and returns the child's exit code.  CPU test with a stand-in rank script."""
import importlib.util
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod_launcher", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_command_line_and_no_launch_cases(monkeypatch):
    b = _bench()
    cmd = b.self_launch_command(["--gpus", "8", "--steps", "5"], 8, port=29611)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5"]
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert b.maybe_self_launch(["--steps", "3"]) is None               # N = 1: nothing to launch
    assert b.maybe_self_launch(["--gpus", "1"]) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert b.maybe_self_launch(["--gpus", "4"]) is None                # already a rank of a launcher


def test_parent_starts_ranks_and_forwards_the_line(tmp_path):
    rank_script = tmp_path / "rank.py"
    rank_script.write_text(textwrap.dedent("""
        import json, os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = __import__("torch").ones(1) * (dist.get_rank() + 1)
        dist.all_reduce(t)
        print("noise from rank", dist.get_rank(), flush=True)
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t), "argv": sys.argv[1:]}), flush=True)
        dist.destroy_process_group()
        sys.exit(3 if "--fail" in sys.argv else 0)
    """))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import importlib.util, sys
        spec = importlib.util.spec_from_file_location("b", {os.path.join(ROOT, "bench.py")!r})
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        sys.exit(b.maybe_self_launch(sys.argv[1:], script={str(rank_script)!r}))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(driver), "--gpus", "2", "--steps", "7"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                       # ONE JSON line on stdout, the ranks' chatter on stderr
    line = json.loads(lines[0])
    assert line == {"n_gpus": 2, "sum": 3.0, "argv": ["--gpus", "2", "--steps", "7"]}
    assert "noise from rank" in p.stderr
    p = subprocess.run([sys.executable, str(driver), "--gpus", "2", "--fail"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert p.returncode != 0                               # the child's failure is the parent's


def test_pick_cpus_follows_the_gpus_numa_node(tmp_path):
    """bench.pick_cpus against a made-up two-socket topology: ranks whose GPUs hang off the same node get consecutive
    blocks of THAT node's CPUs, and a platform that does not expose the topology falls back to block = local rank."""
    import bench
    for node, cpus in ((0, "0-31,64-95"), (1, "32-63,96-127")):
        d = tmp_path / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    node_of = {0: 0, 1: 0, 2: 1, 3: 1, 4: None}.get
    allowed = list(range(128))
    got = [bench.pick_cpus(r, 8, allowed, node_of=node_of, n_gpus=5, sysfs=str(tmp_path)) for r in range(5)]
    assert got[0] == (list(range(0, 8)), 0, "numa")
    assert got[1] == (list(range(8, 16)), 0, "numa")
    assert got[2] == (list(range(32, 40)), 1, "numa")
    assert got[3] == (list(range(40, 48)), 1, "numa")
    assert got[4] == (list(range(32, 40)), None, "block")          # no topology: 8 x local rank over the allowed CPUs
    # an affinity mask that leaves fewer than n CPUs of the node: the block rule again
    cpus, node, rule = bench.pick_cpus(2, 8, list(range(0, 36)), node_of=node_of, n_gpus=5, sysfs=str(tmp_path))
    assert rule == "block" and node == 1 and cpus == list(range(16, 24))
    assert bench._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    # load-aware choice (rule "numa+idle"): node 0 holds two of the GPUs, so each rank chooses inside its half of the node's
    # CPUs; the least busy block wins, a busy SMT sibling counts half, ties go to the first block
    for c in range(128):
        d = tmp_path / "devices" / "system" / "cpu" / f"cpu{c}" / "topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(f"{c % 64},{c % 64 + 64}\n")
    busy = {c: 0.0 for c in range(128)}
    busy.update({0: 0.9, 1: 0.8, 9: 0.3, 64 + 20: 1.0})          # node 0 = 0-31 + 64-95; rank 0's half = CPUs 0-31
    cpus, node, rule = bench.pick_cpus(0, 8, allowed, node_of=node_of, n_gpus=5, sysfs=str(tmp_path), busy=busy)
    assert rule == "numa+idle" and node == 0 and cpus == list(range(10, 18))     # past 0, 1 and 9; 20's sibling is busy
    cpus, node, rule = bench.pick_cpus(1, 8, allowed, node_of=node_of, n_gpus=5, sysfs=str(tmp_path), busy=busy)
    assert rule == "numa+idle" and cpus == list(range(74, 82))     # the other half, 64-95: siblings of 0, 1, 9 and CPU 84 avoided
    idle = {c: 0.0 for c in range(128)}
    assert bench.pick_cpus(0, 8, allowed, node_of=node_of, n_gpus=5, sysfs=str(tmp_path), busy=idle)[0] == list(range(0, 8))
    # /proc/stat sampling: two readings of a made-up file
    stat = tmp_path / "stat"
    stat.write_text("cpu  10 0 10 80 0 0 0 0 0 0\ncpu0 5 0 5 40 0 0 0 0 0 0\ncpu1 5 0 5 40 0 0 0 0 0 0\n")
    assert bench.cpu_busy(sample_s=0.0, proc_stat=str(stat)) == {0: 1.0, 1: 1.0}     # (no time passed: nothing idle)
