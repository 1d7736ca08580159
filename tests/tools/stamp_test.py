import os, sys, ctypes as C, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dd3d_amd import build_model, get_cfg, hip
from dd3d_amd.synthetic import load_calib, make_inputs, make_state_dict
cfg = get_cfg("dd3d_kitti_dla34"); model = build_model(cfg)
model.load_state_dict(make_state_dict(model, calib=load_calib("dla34_kitti"))); model.use_graph = True
inputs = make_inputs(1, 384, 1280)
for _ in range(5): model(inputs)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
hip.lib().dd3d_debug_stamps(buf)
v = list(buf)
print("sort  2..6:", [v[i+1]-v[i] for i in range(2,6)])
print("final 8..11:", [v[i+1]-v[i] for i in range(8,11)], "stage/serial/or:", v[20], v[21], v[22])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
plan = model.get_plan(1, 384, 1280)
e0.record()
for _ in range(20): plan.run()
e1.record(); e1.synchronize()
print("graph ms", e0.elapsed_time(e1)/20)
