import sys, os, ctypes
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libhsa' in l})
if order == 'torch_first':
    import torch
    print('torch avail', torch.cuda.is_available(), torch.version.hip)
    import rdf_fusion_amd as rf
    rf.load_library()
    print(maps())
    s = rf.GpuQuadStore(); print('store ok')
    print('torch avail after', torch.cuda.is_available())
    x = torch.ones(4, device='cuda'); print(x.sum().item())
else:
    import rdf_fusion_amd as rf
    rf.load_library(); print(maps())
    s = rf.GpuQuadStore(); print('store ok')
    import torch
    print(maps())
    print('torch avail', torch.cuda.is_available(), torch.version.hip)
