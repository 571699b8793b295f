"""The RCCL branch of the multi-GPU path on ONE GPU (-m gpu): a process group of backend "nccl" (= RCCL on ROCm) with
world size 1 and the gather forced past the one-rank short-circuit (`force_collective`), so that the code the driver's
2 / 4 / 8-GPU run executes first -- `init_process_group("nccl", device_id=...)`, the asynchronous gather into a kept
receive buffer, the STREAM-side `work.wait()` of `GatherPipeline`, its event-pair timing, the reuse of two message buffers
over many steps -- has run on the hardware before.  Each case is a child process with its own timeout (a hung
rendezvous must not take the test session with it).  No scaling figure is measured here."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def _worker():
    """Child process: GatherPipeline over nccl, world size 1, real collective."""
    import torch
    import torch.distributed as dist

    from quantum_inferno_amd import dist as qdist

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    length, steps, depth = 1 << 20, 9, 2  # 8 MiB messages: the size of one record's reduced product at 2^20 samples
    base = torch.arange(length, dtype=torch.float64, device=dev)
    messages = [torch.empty(length, dtype=torch.float64, device=dev) for _ in range(depth)]
    pipe = qdist.GatherPipeline(depth=depth, dst=0, timing=True, force_collective=True)
    seen = []
    side = torch.cuda.Stream()
    for k in range(steps):
        i = pipe.acquire()  # stream-side wait for the gather that last read messages[i]
        if pipe.outs[i] is not None:
            seen.append((k - depth, pipe.outs[i].clone()))
        # the message is produced by kernels on the compute stream, as a transform's reductions are
        messages[i].copy_(base)
        messages[i].add_(10.0 * k)
        out = pipe.submit(i, messages[i])
        assert out is not None and out.shape == (1, length) and out.data_ptr() != messages[i].data_ptr()
    outs = pipe.drain()
    torch.cuda.synchronize()
    wait_ms = pipe.wait_ms()
    assert len(pipe._pairs) == steps and wait_ms >= 0.0  # one event pair per wait (steps - depth in acquire, depth in drain)
    for k, buf in seen:
        assert torch.equal(buf[0], base + 10.0 * k), k
    for k in (steps - 2, steps - 1):
        assert torch.equal(outs[k % depth][0], base + 10.0 * k), k
    # the receive buffers are kept between calls: one allocation per slot
    assert len(qdist._GATHER_BUFFERS) == depth
    # the synchronous form and the other collectives bench.py issues per run
    got = qdist.gather_reduced(messages[0], force_collective=True)
    assert torch.equal(got[0], messages[0])
    flag = torch.tensor([1.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    every = [torch.zeros_like(flag)]
    dist.all_gather(every, flag)
    dist.barrier()
    torch.cuda.synchronize()
    assert float(every[0].item()) == 1.0
    del side
    qdist.clear_gather_buffers()
    dist.destroy_process_group()
    print(json.dumps({"ok": True, "steps": steps, "wait_ms": wait_ms, "message_bytes": length * 8}))


def test_gather_pipeline_over_rccl_world1():
    run = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=_env(), cwd=ROOT, capture_output=True,
                         text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    rec = json.loads(run.stdout.strip().splitlines()[-1])
    assert rec["ok"] and rec["steps"] == 9


@pytest.mark.parametrize("config,extra", [("1", []), ("2", ["--channels", "4"])])
def test_bench_multirank_code_path_over_rccl_world1(config, extra):
    """bench.py itself with --force-collective 1: the N-rank branch of a leg (two message buffers used in turn, the pipelined
    asynchronous gather per step, settle all-reduce, barriers, the all-gather of the ranks' times) on the real transforms."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, *extra, "--force-collective", "1", "--steps", "6",
           "--warmup", "2", "--cpu-seconds", "0", "--two-streams", "0", "--wrappers", "0"]
    run = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    assert cfg["backend"].startswith("nccl") and cfg["world_size"] == 1 and line["n_gpus"] == 1
    assert cfg["gather_message_bytes_per_rank"] > 0 and line["value"] > 0
    if config == "2":  # (a step of many records: the waits are timed)
        assert cfg["gather_wait_ms_per_step"] is not None and len(cfg["gather_wait_ms_per_step"]) == 1


if __name__ == "__main__":
    if "--worker" in sys.argv:
        _worker()
