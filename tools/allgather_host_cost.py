#!/usr/bin/env python3
"""Host-side cost of one exchange call on the nccl (= RCCL) backend, single rank: what a frame pays in Python /
dispatcher time besides crychic_draw_hot_path (measured on the round-1 box: all_gather(list) 36 us, all_gather_into_tensor
29 us, Draw 27 us)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
buf=torch.zeros((2160,3840,4),dtype=torch.uint8,device="cuda")
views=[buf[0:2160]]
flat=torch.zeros((2160*3840*4,),dtype=torch.uint8,device="cuda")
for name,fn in (("all_gather(list)", lambda: dist.all_gather(views, views[0], async_op=True)),
                ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(flat, buf.view(-1), async_op=True))):
    for _ in range(20): fn().wait()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(500):
        w=fn(); w.wait()
    host=time.perf_counter()-t
    torch.cuda.synchronize(); tot=time.perf_counter()-t
    print("%s: host %.1f us/call, incl. GPU %.1f us/call"%(name,host/500*1e6,tot/500*1e6))
dist.destroy_process_group()
