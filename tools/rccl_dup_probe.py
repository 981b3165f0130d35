import os, sys, torch, torch.distributed as dist
r=int(os.environ['RANK']); w=int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=r, world_size=w, device_id=torch.device('cuda',0))
t=torch.ones(4,device='cuda')*(r+1)
dist.all_reduce(t); torch.cuda.synchronize()
print('rank',r,t.tolist(),flush=True)
dist.destroy_process_group()
