import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests')); sys.path.insert(0, os.path.join(os.getcwd(), 'oracle'))
from __graft_entry__ import load_package
load_package()
import test_host_logic as H
from comfyui_keep_amd.engine import synth
from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
from comfyui_keep_amd.engine.net import KeepNet
from comfyui_keep_amd.modules.keep_processor import KEEPFaceProcessor
from comfyui_keep_amd.modules.keep_model_loader import KEEPModelPack
from comfyui_keep_amd.modules import utils as U
net = KeepNet(**DEFAULT_ARCH); net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
net.to('cuda').eval().set_precision('x3')
pack = KEEPModelPack(net, H._Helper(), None, None, 'KEEP'); pack.device = torch.device('cuda')
proc = KEEPFaceProcessor(pack)
base = synth.ramp_image()
crops = [np.ascontiguousarray(np.roll(base, 17 * k, axis=1)) for k in range(3)]
for gm in ('0', 'auto'):
    net.graph_mode = gm
    dev_faces = proc._restore_crops_u8(crops, 2)
    x = U.crops_to_net_input(crops).unsqueeze(0).cuda()
    outs = net.run_clips([x[:, 0:2], x[:, 2:3]])
    host = [U.net_output_to_bgr_u8(f) for f in torch.cat(outs, 1)[0]]
    # device-converted input vs host-converted input
    u8 = torch.from_numpy(np.stack(crops[:2])).cuda()
    f = torch.empty((2, 512, 512, 3), dtype=torch.float32, device='cuda')
    from comfyui_keep_amd.engine import hiplib as L, ops
    L.call('keep_img2tensor', u8, f, 2 * 512 * 512)
    xd = ops.nhwc_to_nchw(f).view(1, 2, 3, 512, 512)
    print('graph', gm, 'input equal', bool(torch.equal(xd, x[:, 0:2])), 'faces differ:',
          [int((a.astype(np.int16) - b.astype(np.int16) != 0).sum()) for a, b in zip(dev_faces, host)],
          'net(xd) vs net(x):', float((net(xd) - net(x[:, 0:2].contiguous())).abs().max()))
