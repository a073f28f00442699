"""Development helper: does a scene made earlier on the device slow the large-scene render down?
(It did while every scene made its own streams: the later pair shared a hardware queue.)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import raytracingrenderer_amd as rt
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
out = bench.extras(torch, rt, dev, 0)
print("extras alone: large-scene render %.1f ms" % out[2]["render_784x784_256spp"]["ms"], flush=True)
scene = rt.default_scene()
scene.set_device(0)
scene.device_scene()
out = bench.extras(torch, rt, dev, 0)
print("with an idle default scene on the device: %.1f ms" % out[2]["render_784x784_256spp"]["ms"], flush=True)
scene.render(784, 784, 1024)
out = bench.extras(torch, rt, dev, 0)
print("after it rendered to the host: %.1f ms" % out[2]["render_784x784_256spp"]["ms"], flush=True)
