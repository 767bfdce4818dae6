import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l})
if order == "rt_first":
    import raytracer_amd as rt
    ctx = rt.Context(0)
    print("after rt:", maps())
    import torch
    try:
        torch.cuda.set_device(0); print("torch ok", torch.cuda.get_device_name(0))
    except Exception as e:
        print("torch failed:", e)
    print("after torch:", maps())
else:
    import torch
    torch.cuda.set_device(0); print("torch ok", torch.cuda.get_device_name(0))
    print("after torch:", maps())
    import raytracer_amd as rt
    ctx = rt.Context(0)
    print("after rt:", maps())
    s = torch.cuda.ExternalStream(ctx.stream(), device=torch.device("cuda:0"))
    print("ext ok", s)
