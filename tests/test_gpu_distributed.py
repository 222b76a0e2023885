"""GPU side of the multi-GPU path on the one GPU a test box has: the sum buffer
aliased as a torch tensor (no copy) and one RCCL collective on it in a world of
one rank, in a fresh process that loads torch before the library, as bench.py
does.  The partition arithmetic and the two-rank reduce are covered on CPU
(test_distributed_cpu.py, gloo)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_sum_buffer_alias_and_rccl_collective():
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "rccl_single_rank.py")
    out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-2000:] + out.stderr[-4000:]


def test_native_rccl_reduce_in_a_world_of_one():
    """The library's own RCCL route (pt_comm_unique_id / pt_comm_create / pt_comm_create_all / pt_reduce_sum)
    in a process without torch."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "rccl_native_single_rank.py")
    out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-2000:] + out.stderr[-4000:]


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_native_rccl_reduce_in_the_process_shape_of_the_scaling_run():
    """torch.distributed (gloo) first, then the library, then the library's RCCL communicator from an id that
    went through torch.distributed: one HIP runtime, one RCCL, and the reduce leaves the image of a straight render."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "rccl_scaling_process_shape.py")
    out = subprocess.run([sys.executable, script, str(_free_port())], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-3000:] + out.stderr[-4000:]
    print(out.stdout)


def test_bench_under_the_launcher_with_one_rank_uses_the_native_reduce():
    """bench.py as the scaling run starts it (torch.distributed.run, RANK / WORLD_SIZE from the environment) with
    one rank: distributed is off for a world of one, so the line must still be the plain single-GPU line."""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--width", "256", "--height", "256", "--spp-per-step", "8", "--no-cpu-baseline", "--no-extra", "--native-reduce"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["reduce"].startswith("rccl (pt_reduce_sum)")
    assert d["gates"]["samples_per_pixel_root"] == 2 * 8
    assert d["gates"]["primary_hit_ids_bit_exact_vs_reference"] is True


def test_bench_two_rank_rehearsal():
    """bench.py launched the way the scaling run launches it (torch.distributed.run, one process per
    rank), with both ranks on the one GPU and gloo in place of RCCL: the sample partition, the reduce
    into rank 0 and the one JSON line."""
    import json
    import socket

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PT_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--width", "256", "--height", "256", "--spp-per-step", "4", "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "Mpaths/s"
    assert d["config"]["paths_per_step"] == 256 * 256 * 4 * 2
    assert d["gates"]["samples_per_pixel_root"] == 3 * 4 * 2  # both ranks' samples arrived at the root
    assert d["gates"]["primary_hit_ids_bit_exact_vs_reference"] is True
    assert d["value"] > 0 and "roofline" in d and "cpu_baseline" not in d
