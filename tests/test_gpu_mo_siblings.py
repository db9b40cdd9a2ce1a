"""PixelNet's MOGRU4Rec and MONextItNet (code/REC/model/PixelNet/{mogru4rec,monextitnet}.py) on the GPU: the visual encoder in
front of the GRU / NextItNet blocks.  Oracle = the pieces that are each pinned against the reference on their own, composed the
way the reference's MO* classes compose them (`.view(B, -1, 2, D)`, pos = [:, :, 0], neg = [:, :, 1] -- the same lines as
MOSASRec, whose golden pins that composition): HF CLIPVisionModel + MeanItemEncoder (oracle/mosasrec_oracle.py) -> the GRU
(oracle/gru4rec_oracle.py) or residual-block (oracle/nextitnet_oracle.py) restatement."""
import numpy as np
import pytest
import torch

from oracle import gru4rec_oracle as GO
from oracle import mosasrec_oracle as MO
from oracle import nextitnet_oracle as NO

pytestmark = pytest.mark.gpu

D, L, B = 64, 6, 3
TUNE = 5 + 16 * 2            # tiny tower: train block 2 + rec_fc


def _config(extra):
    cfg = {"embedding_size": D, "initializer_range": 0.02, "MAX_ITEM_LIST_LENGTH": L, "seed": 2020,
           "encoder_name": "clip-vit-tiny-test", "encoder_source": "transformers", "pretrain_path": None,
           "fine_tune_arg": {"tune_scale": TUNE, "pre_trained": True, "activation": "relu", "dnn_layers": [], "method": "mean"}}
    cfg.update(extra)
    return cfg


CASES = {
    "MOGRU4Rec": ({"hidden_size": 2, "num_layers": 2, "dropout_prob": 0.0},
                  lambda p, emb, mask: GO.forward_loss_rows(p, emb, mask, 2)),
    "MONextItNet": ({"block_num": 2, "dilations": [1, 2], "kernel_size": 3, "final_layer": True},
                    lambda p, emb, mask: NO.forward_loss_rows(p, emb, mask, [1, 2, 1, 2])),
}


@pytest.mark.parametrize("name", list(CASES))
def test_mo_sibling_loss_and_gradients_match_the_composed_oracle(name):
    import pixelrec_amd.model as M
    from pixelrec_amd.model.visual import ENCODER_SHAPES

    extra, oracle_loss = CASES[name]

    class DL:
        item_num = 50

    torch.manual_seed(1)
    hf = MO.hf_clip_vision(*ENCODER_SHAPES["clip-vit-tiny-test"])
    m = getattr(M, name)(_config(extra), DL())
    m.visual_encoder.item_encoder.load_state_dict(
        {k: v for k, v in MO.hf_state_to_reference_names(hf).items() if "post_layernorm" not in k}, strict=True)
    names = [n for n, _ in m.named_parameters()]
    assert names[0].startswith("visual_encoder.") and not any(n.startswith("item_embedding") for n in names)
    block = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith("visual_encoder")}
    m = m.cuda().train()
    g = torch.Generator().manual_seed(0)
    images = torch.randn(B, 2 * (L + 1), 3, 64, 64, generator=g)
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[0, :3] = 0
    images[0, :8] = 0.0
    loss = m((images.cuda(), mask.cuda()))
    loss.backward()

    rec_w = m.visual_encoder.rec_fc[0].weight.detach().cpu().clone().requires_grad_(True)
    rec_b = m.visual_encoder.rec_fc[0].bias.detach().cpu().clone().requires_grad_(True)
    hf.train(False)
    sp = {k: v.clone().requires_grad_(True) for k, v in block.items()}
    emb = MO.mean_item_encoder(hf, rec_w, rec_b, images.flatten(0, 1)).view(B, -1, 2, D)
    ref = oracle_loss(sp, emb, mask)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 3e-5 * max(1.0, abs(float(ref.detach())))
    m.join_weight_grads()
    checked = 0
    for k, v in m.named_parameters():
        if k.startswith("visual_encoder"):
            continue
        err = (v.grad.cpu() - sp[k].grad).abs().max().item()
        assert err <= 5e-6 + 3e-4 * sp[k].grad.abs().max().item(), (k, err)
        checked += 1
    assert checked == len(block)
    # the gradient w.r.t. the encoder output reaches the encoder's trainable tail (pxr_mosasrec_emb_grad_f32 + tower backward)
    assert (m.visual_encoder.rec_fc[0].weight.grad.cpu() - rec_w.grad).abs().max().item() < 1e-5 + 3e-4 * rec_w.grad.abs().max().item()
    # inference: compute_item + predict
    m.eval()
    with torch.no_grad():
        item_imgs = torch.randn(20, 3, 64, 64, generator=g)
        feat = m.compute_item(item_imgs.cuda())
        ref_feat = MO.mean_item_encoder(hf, rec_w, rec_b, item_imgs)
        assert (feat.cpu() - ref_feat).abs().max().item() < 2e-5 * max(1.0, ref_feat.abs().max().item())
        seq = torch.randint(1, 20, (4, L), generator=g)
        scores = m.predict(seq.cuda(), feat)
        if name == "MOGRU4Rec":
            want = GO.predict({**{k: v.detach() for k, v in sp.items()}}, seq, ref_feat, 2)
        else:
            want = NO.encode_rows({k: v.detach() for k, v in sp.items()}, ref_feat[seq], [1, 2, 1, 2])[:, -1] @ ref_feat.t()
        assert (scores.cpu() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("name,model_yaml", [
    ("MOGRU4Rec", "model: MOGRU4Rec\nnum_layers: 1\nembedding_size: 32\nhidden_size: 1\ndropout_prob: 0\ninitializer_range: 0.02\n"),
    ("MONextItNet", "model: MONextItNet\nembedding_size: 32\nkernel_size: 3\nblock_num: 2\ndilations: [1,4]\nfinal_layer: False\n"),
])
def test_mo_sibling_trainer_end_to_end(tmp_path, name, model_yaml):
    """PixelNet/{gru4rec,nextitnet}.yaml-shaped runs through the main.py surface: data -> model -> Trainer.fit / evaluate with
    the HBM-resident image store, two optimizer groups, full-sort evaluation over encoder-produced item features."""
    from pixelrec_amd.config import Config
    from pixelrec_amd.data import bulid_dataloader, load_data
    from pixelrec_amd.optim import OptimizerGroup
    from pixelrec_amd.parallel import DataParallel
    from pixelrec_amd.trainer import Trainer
    from pixelrec_amd.utils import get_model
    from tests.golden_util import GOLDEN_DIR

    my, ov = tmp_path / "m.yaml", tmp_path / "o.yaml"
    my.write_text(model_yaml)
    ov.write_text(f"seed: 2020\nstate: INFO\nuse_modality: True\nreproducibility: True\ncheckpoint_dir: '{tmp_path}/saved'\n"
                  f"log_path: '{tmp_path}/log'\nshow_progress: False\nMAX_ITEM_LIST_LENGTH: 6\ndata_path: {GOLDEN_DIR}/\n"
                  "dataset: TinyInter\nimage_path: 'synthetic:64'\nencoder_name: 'clip-vit-tiny-test'\n"
                  "encoder_source: 'transformers'\nepochs: 3\ntrain_batch_size: 8\n"
                  "fine_tune_arg: {tune_scale: 37, pre_trained: True, activation: 'relu', dnn_layers: [], method: 'mean'}\n"
                  "optim_args: {modal_lr: 0.001, rec_lr: 0.001, modal_decay: 0, rec_decay: 0.1}\n"
                  "eval_batch_size: 16\ntopk: [5,10]\nmetrics: ['Recall', 'NDCG']\nvalid_metric: NDCG@10\n"
                  "metric_decimal_place: 7\neval_step: 1\nstopping_step: 30\n")
    config = Config([str(my), str(ov)])
    config["device"] = torch.device("cuda", 0)
    dataload = load_data(config)
    train, valid, test = bulid_dataloader(config, dataload)
    model = get_model(config["model"])(config, dataload).to(config["device"])
    assert type(model).__name__ == name
    trainer = Trainer(config, DataParallel(model))
    assert isinstance(trainer.optimizer, OptimizerGroup)
    trainer.fit(train, valid, saved=True)
    losses = [trainer.train_loss_dict[e] for e in sorted(trainer.train_loss_dict)]
    assert len(losses) == 3 and losses[-1] < losses[0]
    out = trainer.evaluate(test, load_best_model=True)
    assert set(out) == {"recall@5", "recall@10", "ndcg@5", "ndcg@10"}
    assert trainer.item_feature.shape == (dataload.item_num, 32)
