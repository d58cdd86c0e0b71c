import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29733")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
for n in (20_000_000, 60_000_000, 70_000_000, 100_000_000, 200_000_000):
    for shape in ("2d", "flat"):
        src = torch.arange(2*n, dtype=torch.int64, device="cuda")
        send = src.view(n,2) if shape=="2d" else src
        recv = torch.zeros_like(send)
        cnt = [n] if shape=="2d" else [2*n]
        dist.all_to_all_single(recv, send, output_split_sizes=cnt, input_split_sizes=cnt)
        torch.cuda.synchronize()
        bad = (recv.view(-1) != src).sum().item()
        first_bad = int((recv.view(-1) != src).nonzero()[0]) if bad else -1
        print(n, shape, "bytes", 16*n, "mismatches", bad, "first", first_bad, flush=True)
dist.destroy_process_group()
