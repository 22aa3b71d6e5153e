"""RCCL probe for a GPU box: ufd_create_replicas (dlopen'd librccl) with NCCL_DEBUG=INFO; then torch.distributed nccl, world 1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NCCL_DEBUG", "INFO")
which = sys.argv[1] if len(sys.argv) > 1 else "ufd"
if which == "ufd":
    from infercam_onnx_amd import nn, synth

    ms = nn.UltrafaceModel.create_replicas(nn.UltrafaceVariant.W320H240, 0.5, 0.5, [0], weights=synth.synthetic_weights())
    print("replicas ok", ms[0].placement())
    with open("/proc/self/maps") as f:
        print(sorted({l.split()[-1] for l in f if "rccl" in l or "amdhip" in l}))
else:
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda")
    dist.broadcast(t, 0)
    torch.cuda.synchronize()
    print("torch nccl ok", t.cpu())
    dist.destroy_process_group()
