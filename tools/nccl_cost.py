import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
flat = torch.empty(2, dtype=torch.int64, device=dev)
def verify():
    mine = torch.tensor([123, 456], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(flat, mine)
    return flat.tolist()
pin = torch.empty(2, dtype=torch.int64).pin_memory()
mine2 = torch.empty(2, dtype=torch.int64, device=dev)
def verify2():
    pin[0] = 123; pin[1] = 456
    mine2.copy_(pin, non_blocking=True)
    dist.all_gather_into_tensor(flat, mine2)
    return flat.tolist()
for name, fn in (("tensor()+all_gather+tolist", verify), ("pinned copy+all_gather+tolist", verify2)):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): fn()
    print("%s: %.1f us per call" % (name, (time.perf_counter() - t0) / 300 * 1e6))
x = torch.zeros(1024, dtype=torch.uint8, device=dev)
def ag_only():
    dist.all_gather_into_tensor(flat, mine2)
for _ in range(20): ag_only()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): ag_only()
torch.cuda.synchronize()
print("all_gather alone (no sync each): %.1f us per call" % ((time.perf_counter() - t0) / 300 * 1e6))
dist.destroy_process_group()
