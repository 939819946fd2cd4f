"""Where does the posterior error come from?  (developer tool; needs a GPU)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import matsed_oracle as O
from transformer4sed_amd import synth
from transformer4sed_amd.passt_sed import PaSST_SED
from transformer4sed_amd.engine import SedEngine
from transformer4sed_amd.ops import call

dev = "cuda"
depth = 2
net = PaSST_SED(passt_feature_layer=depth, f_pool="mean_pool", decode_ratio=10, at_adapter=True, decoder="transformerXL",
                decoder_layer_num=3, mlm=False, load_pretrained_model=False, encoder_depth=depth)
sd = synth.matsed_state_dict_np(tag="w768", depth=12)
own = net.state_dict()
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k in own}, strict=True)
net = net.to(dev).eval()
mel = torch.from_numpy(synth.det_uniform("model_d768_l2/mel", (2, 128, 1000), -1.2, 1.2))
with torch.no_grad():
    ref = O.passt_sed_forward(O.to_torch_sd(sd), mel, depth=depth, feature_layer=depth)
    s, w, o = net(mel.to(dev))
print("full strong err", float((s.cpu() - ref["strong"]).abs().max()))
fbm_err = (o["frame_before_mask"].cpu() - ref["frame_before_mask"]).abs()
print("frame_before_mask err max/mean", float(fbm_err.max()), float(fbm_err.mean()), "scale", float(ref["frame_before_mask"].abs().mean()))
eng = net.engine
W = eng._weights(False)


def head(xd, temp=1.0):
    B, T, _ = xd.shape
    strong = torch.empty(B, 10, T, device=dev); weak = torch.empty(B, 10, device=dev); sums = torch.empty(B, 10, 2, device=dev)
    call("sed_head_fwd", xd, eng.P("classifier.weight"), eng.P("classifier.bias"), temp, None, strong, weak, sums, B, T, 10)
    return strong


with torch.no_grad():
    xd, _ = eng._decoder_fwd(W, ref["frame_before_mask"].to(dev).contiguous(), False)
    print("decoder-only (exact input) strong err", float((head(xd).cpu() - ref["strong"]).abs().max()))
    for li in range(3):
        pass
    xe = (xd.cpu() - ref["decoder_out"]).abs()
    print("decoder out err max/mean", float(xe.max()), float(xe.mean()), "scale", float(ref["decoder_out"].abs().mean()))
    # head on exact decoder output
    print("head-only err", float((head(ref["decoder_out"].to(dev).contiguous()).cpu() - ref["strong"]).abs().max()))
    # decoder fed with HIP frame_before_mask but compared with oracle decoder run on the same (HIP) input: isolates decoder
    ref2 = O.context_net(O.to_torch_sd(sd), o["frame_before_mask"].cpu(), 3)
    xd2, _ = eng._decoder_fwd(W, o["frame_before_mask"].contiguous(), False)
    print("decoder err on HIP input", float((xd2.cpu() - ref2).abs().max()), "mean", float((xd2.cpu() - ref2).abs().mean()))
