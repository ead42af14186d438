"""Does the per-frame CNN (DepthFusionNet on MIOpen) repeat bit for bit when the module's caches are reset and rebuilt?  (the repeatability bars of
tests/test_dropin_module.py sit on this.)  python tools/miopen_repeat.py [loops] [deterministic 0/1]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_cases import build_setup_case
from tests.test_dropin_module import _module_and_data
loops = int(sys.argv[1]) if len(sys.argv) > 1 else 30
det = len(sys.argv) > 2 and sys.argv[2] == "1"
if det:
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
dev = torch.device("cuda:0")
worst_maps = worst_rgb = 0.0
nbit = 0
for case_name in ("setup", "setup_holes"):
    net, data, rd = _module_and_data(build_setup_case(case_name), dev, "f16mx")
    rd["depth_range"] = data["depth_range"][0]
    ref_maps = ref_rgb = None
    for it in range(loops):
        net.support_neural_points = None
        net.multiview_aggregator.vis_featmaps = None
        with torch.no_grad():
            out = net.render_rays(data, rd)
        maps = net.multiview_aggregator.vis_featmaps.detach().clone()
        if ref_maps is None:
            ref_maps, ref_rgb = maps, out["rgb"].clone()
            continue
        em = float((maps - ref_maps).abs().max() / ref_maps.abs().max())
        er = float((out["rgb"] - ref_rgb).abs().max() / ref_rgb.abs().max())
        worst_maps, worst_rgb = max(worst_maps, em), max(worst_rgb, er)
        nbit += int(em == 0.0)
print(f"deterministic={det}: {2 * (loops - 1)} rebuilds, {nbit} bit-identical maps; worst relative difference of the maps {worst_maps:.3e}, of rgb {worst_rgb:.3e}")
