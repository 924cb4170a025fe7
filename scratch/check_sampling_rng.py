import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, seeded
from lvt_amd.config import get_cfg
from lvt_amd.modeling import build_model
cfg = get_cfg(); cfg.merge_from_file("configs/vt/DSFVT.yaml"); cfg.MODEL.DEVICE = "cuda"; cfg.OUTPUT_DIR = "/tmp/x"
cfg.TEST.EVALUATORS = "VTSampler"
m = build_model(cfg).eval()
m.model.load_state_dict(seeded.seeded_params(seeded.dsfvt_shapes(), 4321), strict=False)
codes = torch.stack([seeded.seeded_codes("v%d" % i, (16, 4, 16, 16), 8) for i in range(2)]).transpose(1, 2).contiguous().cuda()
torch.manual_seed(0)
a = m.sample_video(codes.clone(), n_prime=14)
b = m.sample_video(codes.clone(), n_prime=14)          # graphs replayed
torch.manual_seed(0)
c = m.sample_video(codes.clone(), n_prime=14)
print("primed frames untouched:", bool((a[:, :, :14] == codes[:, :, :14]).all()))
print("run1 vs run2 differ (fresh randomness on replay):", float((a[:, :, 14:] != b[:, :, 14:]).float().mean()))
print("same seed reproduces:", float((a[:, :, 14:] == c[:, :, 14:]).float().mean()))
print("distinct codes in generated frames:", int(a[:, :, 14:].unique().numel()), "sample0 vs sample1 differ:", float((a[0, :, 14:] != a[1, :, 14:]).float().mean()))
