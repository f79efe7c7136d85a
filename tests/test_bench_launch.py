"""`python bench.py --gpus N` without a launcher: the parent starts the N ranks itself (bench.launch_ranks).  CPU tests of
the spawn / aggregate / failure logic with stand-in rank scripts (the real ranks need GPUs); the one-GPU rehearsal of the
real thing is tests/test_parallel.py::test_bench_self_launch_two_ranks_on_one_gpu."""
import json
import os
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launch_ranks_sets_the_rendezvous_environment_and_forwards_rank0(tmp_path):
    import bench
    script = _script(tmp_path, """
        import json, os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_WORLD_SIZE"] == str(w)
        open(os.path.join(sys.argv[1], f"seen{r}"), "w").write(os.environ["MASTER_PORT"])
        print("noise before the line")
        if r == 0:
            print(json.dumps({"n_gpus": w, "argv": sys.argv[2:]}))
    """)
    rc, out = bench.launch_ranks(3, [str(tmp_path), "--gpus", "3", "--steps", "2"], script=script, timeout_s=60)
    assert rc == 0
    line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"n_gpus": 3, "argv": ["--gpus", "3", "--steps", "2"]}
    ports = {open(tmp_path / f"seen{r}").read() for r in range(3)}
    assert len(ports) == 1  # every rank got the same rendezvous port


def test_launch_ranks_fails_fast_when_one_rank_dies(tmp_path):
    import bench
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)
        time.sleep(600)   # a rank waiting in a collective for the dead one
    """)
    t0 = time.time()
    rc, out = bench.launch_ranks(2, [], script=script, timeout_s=120)
    assert rc == 3 and time.time() - t0 < 60


def test_launch_ranks_times_out(tmp_path):
    import bench
    script = _script(tmp_path, "import time; time.sleep(600)")
    rc, _ = bench.launch_ranks(2, [], script=script, timeout_s=1.0)
    assert rc == 124


def test_gpus_flag_disagreeing_with_a_launcher_is_an_error(tmp_path, monkeypatch):
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "launcher started 2 ranks" in r.stderr
