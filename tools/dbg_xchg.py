#!/usr/bin/env python3
"""Dev probe: two ranks on cuda:0 exchanging gradients through xgmi.OneShotExchange; prints the error word and the time of
every phase (set-up, self test, eager steps, capture, replays)."""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def main(rank, world, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hcatgnet_amd as H
    from hcatgnet_amd import synth
    from hcatgnet_amd.ddp import DataParallelGCN
    from hcatgnet_amd.xgmi import OneShotExchange
    t0 = time.perf_counter()
    def say(msg):
        print(f"[rank {rank} +{time.perf_counter() - t0:6.2f}s] {msg}", flush=True)
    sb = synth.make_config("C2", num_graphs=256, rank=rank)
    batch = sb.as_batch("cuda")
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    dp = DataParallelGCN(m, combine="sse")
    xchg = OneShotExchange(sum(p.numel() for p in m.parameters()))
    say(f"setup ok={xchg.ok}")
    say(f"selftest {xchg.self_test()}")
    step = xchg.attach(dp.make_train_step())
    if os.environ.get("MEET", "1") == "1":
        def meet():
            torch.cuda.synchronize(); dist.barrier()
        step.pre_exchange_hook = meet
    for i in range(5):
        loss = float(step(batch))
        say(f"eager step {i}: loss {loss:.5f} err {int(xchg.err[0])} opt step {m.optimizer.steps_done()}")
    if os.environ.get("MEET", "1") == "1":
        dist.barrier(); xchg.close(); dist.destroy_process_group(); return
    step.capture(batch)
    torch.cuda.synchronize()
    say(f"captured: err {int(xchg.err[0])} opt step {m.optimizer.steps_done()}")
    for i in range(5):
        loss = float(step.replay())
        say(f"replay {i}: loss {loss:.5f} err {int(xchg.err[0])} opt step {m.optimizer.steps_done()}")
    for i in range(200):
        step.replay()
    torch.cuda.synchronize()
    say(f"200 replays back to back: err {int(xchg.err[0])} loss {float(step.replay()):.5f}")
    dist.barrier()
    xchg.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(main, args=(2, port), nprocs=2, join=True)
