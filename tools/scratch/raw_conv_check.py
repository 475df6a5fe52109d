import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from tests.helpers import build_model
from shasta_amd import shared_conv_train as sct
dev = torch.device("cuda", 0)
for (B, cin, H, W) in [(3, 40, 33, 47), (3, 48, 33, 47), (2, 8, 24, 24), (1, 16, 12, 200), (2, 64, 33, 47), (1, 48, 7, 5)]:
    for arith in ("f16x2", "f32"):
        m = build_model(dict(max_obj=4, np=1, nf=3, seed=3, cin=cin, stride=8)).to(dev)
        m.arithmetic = arith
        g = torch.Generator().manual_seed(1)
        x = torch.relu(torch.randn(B, cin, H, W, generator=g)).to(dev)
        xp = torch.relu(torch.randn(B, cin, H, W, generator=g)).to(dev)
        y, yp, xmax = sct._raw_conv(m, x, xp)
        conv = m.shared_conv[0]
        ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1).permute(0, 2, 3, 1)
        refp = F.conv2d(xp.double(), conv.weight.double(), conv.bias.double(), padding=1).permute(0, 2, 3, 1)
        m.eval()
        with torch.no_grad():
            o, op = m.shared_conv_nhwc(x, xp)
            oref = m.shared_conv(x).permute(0, 2, 3, 1)
        print(B, cin, H, W, arith, "raw err %.3e %.3e" % (float((y - ref).abs().max()), float((yp - refp).abs().max())),
              "eval err %.3e" % float((o - oref).abs().max()), "xmax", xmax.view(torch.float32).tolist()[:2], float(x.abs().amax()))
