"""developer aid: which combination of (library RCCL use, torch import) upsets process exit"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
mode = sys.argv[1]
sc = scenes.make_scene(seed=314, N=3000, Nmap=60000, L=3)
if "torch_first" in mode:
    import torch
    torch.zeros(4, device="cuda")
if "node" in mode:
    nd = capi.Node(sc["params"], [0], exchange=capi.XCHG_RCCL if "rccl" in mode else capi.XCHG_HOST)
    nd.map_build(sc["map"]); nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    print(nd.measure(sc["state0"], True)["M"])
    nd.close()
if "torch_after" in mode:
    import torch
    torch.zeros(4, device="cuda")
    if "staged" in mode:
        from malio_amd import dist as mdist
print("done", mode, flush=True)
