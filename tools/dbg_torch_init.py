import subprocess, sys
BASE = '''
import sys, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=1)
e = capi.Engine(sc["params"])
e.map_build(sc["map"])
%s
import torch
print("torch sees", torch.cuda.device_count(), flush=True)
torch.zeros(4, device="cuda")
print("OK")
'''
VARIANTS = {
    "build_only": "",
    "add_ds": "print(e.map_add(sc['map'][:1000], True))",
    "add_nods": "print(e.map_add(sc['map'][:1000], False))",
    "delete": "print(e.map_delete_boxes(np.array([[0,0,0,5,5,5]], np.float32)))",
    "get": "print(e.map_get().shape)",
    "nearest": "print(e.nearest_search(sc['map'][:10], 5)[2])",
}
for k, v in VARIANTS.items():
    r = subprocess.run([sys.executable, "-c", BASE % v], capture_output=True, text=True)
    print("==", k, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], "|", (r.stderr.strip().splitlines() or [""])[-1][:200])
