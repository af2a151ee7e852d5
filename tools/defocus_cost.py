import importlib, sys
sys.path.insert(0, '.')
srt = importlib.import_module("cuda-spectral-ray-tracer_amd")
scene = srt.Scene.builtin(100, 0).build_bvh(1, 1984)
W, H = 1920, 1080
r = srt.Renderer(0); r.upload_scene(scene); r.set_partition(0, 1)
for label, da in (("defocus 0.6 (default)", None), ("defocus 0", 0.0), ("defocus 0.6 (default)", None), ("defocus 0", 0.0)):
    cam = scene.default_camera(W, H)
    if da is not None: cam.defocus_angle = da
    r.set_camera(cam)
    best = 1e9
    for _ in range(2):
        r.init_device_params(W, H, 256, 16, 1984); r.render_chunk(W, H); r.synchronize(); best = min(best, r.last_kernel_ms())
    print(label, round(best, 2), "ms", r.stats()["rays"])
