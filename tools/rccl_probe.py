"""Probe (development aid): can two RCCL ranks share one GPU on this box?  If so the N > 1 path of bench.py can be smoke-tested
on a 1-GPU box (QOA_BENCH_SHARE_GPU=1 maps every LOCAL_RANK to device 0)."""
import os, sys, subprocess, socket
if "RANK" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    rc = [p.wait(timeout=120) for p in ps]
    print("exit codes", rc)
    sys.exit(max(rc))
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0))
x = torch.full((4,), float(rank + 1), device="cuda:0")
out = [torch.empty(4, device="cuda:0") for _ in range(2)] if rank == 0 else None
dist.gather(x, gather_list=out, dst=0)
torch.cuda.synchronize()
if rank == 0:
    print("gathered", [o.tolist() for o in out])
dist.barrier()
dist.destroy_process_group()
