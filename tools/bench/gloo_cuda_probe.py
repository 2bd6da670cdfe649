"""Can two ranks share ONE GPU through gloo with device tensors?  (RCCL refuses two ranks per device; this is for validating the
world_size > 1 host logic with the real HIP kernels on a 1-GPU box.)"""
import os, sys, torch, torch.distributed as td
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
td.init_process_group("gloo")
dev = torch.device("cuda", 0)
for dt in (torch.float32, torch.bfloat16):
    send = torch.full((4, 8), float(rank + 1), device=dev, dtype=dt)
    recv = torch.zeros(world, 4, 8, device=dev, dtype=dt)
    try:
        td.all_gather_into_tensor(recv.view(-1), send.view(-1))
        print(rank, dt, "all_gather_into_tensor OK", recv[:, 0, 0].tolist(), flush=True)
    except Exception as ex:
        print(rank, dt, "all_gather_into_tensor FAILED:", type(ex).__name__, str(ex)[:200], flush=True)
        lst = [torch.zeros_like(send) for _ in range(world)]
        try:
            td.all_gather(lst, send)
            print(rank, dt, "all_gather(list) OK", [float(t[0, 0]) for t in lst], flush=True)
        except Exception as ex2:
            print(rank, dt, "all_gather(list) FAILED:", type(ex2).__name__, str(ex2)[:200], flush=True)
t = torch.tensor([float(rank)], device=dev, dtype=torch.float64)
try:
    td.all_reduce(t, op=td.ReduceOp.MAX); print(rank, "all_reduce OK", float(t), flush=True)
except Exception as ex:
    print(rank, "all_reduce FAILED", str(ex)[:200], flush=True)
g2 = td.new_group(ranks=list(range(world)))
td.barrier()
td.destroy_process_group()
