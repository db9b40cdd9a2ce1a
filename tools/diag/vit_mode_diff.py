"""Diagnostic: parameter error of the native tower after 3 VisualAdamW steps vs torch (tests/test_gpu_vit.py scenario),
in both GEMM modes."""
import copy, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
from pixelrec_amd.model import visual
from pixelrec_amd.optim import VisualAdamW

for mode in ("f32", "bf16x3"):
    ops.set_gemm_mode(mode)
    torch.manual_seed(5)
    cfg = {"encoder_name": "clip-vit-tiny-test", "encoder_source": "transformers", "embedding_size": 24, "pretrain_path": None,
           "fine_tune_arg": {"tune_scale": 0, "pre_trained": False, "activation": "relu", "dnn_layers": [], "method": "mean"}}
    enc = visual.load_model(cfg)
    for p in enc.parameters():
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    ref = copy.deepcopy(enc)
    enc = enc.cuda()
    opt = VisualAdamW(enc, lr=1e-2, weight_decay=0.05, eps=1e-2)
    topt = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05, eps=1e-2)
    for step in range(3):
        x = torch.randn(4, 3, 64, 64); w = torch.randn(4, 24) * 1e4
        out = enc(x.cuda()); (out * w.cuda()).sum().backward()
        tower = ref.item_encoder(x)[0]
        ref_out = torch.mean(ref.rec_fc(tower), dim=1)
        topt.zero_grad(); (ref_out * w).sum().backward()
        worst = max(((p.grad.cpu() - q.grad).abs().max().item() / max(q.grad.abs().max().item(), 1e-30), n)
                    for (n, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()) if q.grad is not None)
        print(mode, "step", step, "worst rel grad err", worst)
        opt.step(); topt.step()
    errs = sorted(((p.detach().cpu() - q).abs().max().item(), n) for (n, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()))
    print(mode, "worst param errs", errs[-4:])
