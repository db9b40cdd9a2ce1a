"""PixelNet path on the GPU: MOSASRec (visual encoder + shared sequence block) against the CPU oracle built on HF's
CLIPVisionModel (parity for the ViT part is anchored on the HF class, see oracle/mosasrec_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import mosasrec_oracle as MO
from oracle import sasrec_oracle as O

pytestmark = pytest.mark.gpu

D, L, H, NL, B = 64, 6, 2, 2, 3
TUNE = 5 + 16 * 2            # tiny tower has 3 blocks: freeze embeddings + blocks 0,1 ; train block 2 (+ rec_fc)


def _config(p_drop=0.0):
    return {"n_layers": NL, "n_heads": H, "embedding_size": D, "inner_size": 2, "hidden_dropout_prob": p_drop,
            "attn_dropout_prob": p_drop, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
            "MAX_ITEM_LIST_LENGTH": L, "seed": 2020, "encoder_name": "clip-vit-tiny-test",
            "encoder_source": "transformers", "pretrain_path": None,
            "fine_tune_arg": {"tune_scale": TUNE, "pre_trained": True, "activation": "relu", "dnn_layers": [], "method": "mean"}}


def _build():
    from pixelrec_amd.model import MOSASRec
    from pixelrec_amd.model.visual import ENCODER_SHAPES

    class DL:
        item_num = 50

    torch.manual_seed(1)
    hf = MO.hf_clip_vision(*ENCODER_SHAPES["clip-vit-tiny-test"])
    m = MOSASRec(_config(), DL())
    m.visual_encoder.item_encoder.load_state_dict(
        {k: v for k, v in MO.hf_state_to_reference_names(hf).items() if "post_layernorm" not in k}, strict=True)
    seq = {k: v for k, v in O.synth_params(50, D, L, NL, 2, seed=4).items() if k != "item_embedding.weight"}
    m.load_state_dict(seq, strict=False)
    return m, hf, seq


def test_parameter_groups_and_frozen_prefix():
    m, hf, seq = _build()
    names = [n for n, p in m.named_parameters()]
    assert any(n.startswith("visual_encoder.item_encoder.vision_model.") for n in names)      # reference key layout
    assert "visual_encoder.rec_fc.0.weight" in names
    frozen = [n for n, p in m.visual_encoder.item_encoder.named_parameters() if not p.requires_grad]
    assert len(frozen) == TUNE and all("layers.2." not in n for n in frozen)


def test_loss_and_gradients_match_oracle():
    m, hf, seq = _build()
    m = m.cuda().train()
    g = torch.Generator().manual_seed(0)
    images = torch.randn(B, 2 * (L + 1), 3, 64, 64, generator=g)
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[0, :3] = 0
    images[0, :8] = 0.0                                   # left-padded sequence: zero images, masked targets
    loss = m((images.cuda(), mask.cuda()))
    loss.backward()

    # ---- oracle: HF tower + rec_fc (mean) -> interleaved item_emb -> pinned sequence oracle
    rec_w = m.visual_encoder.rec_fc[0].weight.detach().cpu().clone().requires_grad_(True)
    rec_b = m.visual_encoder.rec_fc[0].bias.detach().cpu().clone().requires_grad_(True)
    for p in hf.parameters():
        p.requires_grad_(True)
    hf.train(False)
    sp = {k: v.clone().requires_grad_(True) for k, v in seq.items()}
    emb = MO.mean_item_encoder(hf, rec_w, rec_b, images.flatten(0, 1)).view(B, -1, 2, D)
    cfg = {"n_layers": NL, "n_heads": H, "layer_norm_eps": 1e-12}
    ref = MO.forward_loss(sp, emb, mask, cfg)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref)) < 3e-5 * max(1.0, abs(float(ref)))
    # sequence-block parameters
    for k, v in m.named_parameters():
        if k.startswith("visual_encoder"):
            continue
        err = (v.grad.cpu() - sp[k].grad).abs().max().item()
        assert err <= 5e-6 + 3e-4 * sp[k].grad.abs().max().item(), (k, err)
    # trainable tail of the visual encoder (gradient arrives through pxr_mosasrec_emb_grad_f32 + torch autograd)
    assert (m.visual_encoder.rec_fc[0].weight.grad.cpu() - rec_w.grad).abs().max().item() < 1e-5 + 3e-4 * rec_w.grad.abs().max().item()
    assert (m.visual_encoder.rec_fc[0].bias.grad.cpu() - rec_b.grad).abs().max().item() < 1e-5 + 3e-4 * rec_b.grad.abs().max().item()
    hf_named = dict(hf.named_parameters())
    checked = 0
    for n, p in m.visual_encoder.item_encoder.named_parameters():
        ref_p = hf_named[n[len("vision_model."):]]
        if p.requires_grad and ref_p.grad is None:       # post_layernorm: trainable but unused by method 'mean'
            assert p.grad is None
        elif p.requires_grad:
            err = (p.grad.cpu() - ref_p.grad).abs().max().item()
            assert err <= 1e-6 + 5e-4 * ref_p.grad.abs().max().item(), (n, err)
            checked += 1
        else:
            assert p.grad is None
    assert checked == 16


def test_predict_and_compute_item():
    m, hf, seq = _build()
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn(50, 3, 64, 64, generator=g)
    feat = m.compute_item(imgs.cuda())
    rec_w = m.visual_encoder.rec_fc[0].weight.detach().cpu()
    rec_b = m.visual_encoder.rec_fc[0].bias.detach().cpu()
    with torch.no_grad():
        ref_feat = MO.mean_item_encoder(hf, rec_w, rec_b, imgs)
    assert (feat.cpu() - ref_feat).abs().max().item() < 3e-5
    item_seq = torch.randint(1, 50, (4, L), generator=g)
    item_seq[0, :2] = 0
    scores = m.predict(item_seq.cuda(), feat).cpu()
    p = dict(seq)
    p["item_embedding.weight"] = ref_feat
    ref = O.predict(p, item_seq, ref_feat, {"n_layers": NL, "n_heads": H, "layer_norm_eps": 1e-12})
    assert (scores - ref).abs().max().item() < 1e-4


def test_image_transform_kernel():
    from pixelrec_amd import ops

    rng = np.random.default_rng(0)
    store = rng.integers(0, 256, size=(7, 16, 12, 3), dtype=np.uint8)
    ids = np.array([[3, 0, 6], [1, 1, 5]], dtype=np.int64)
    out = ops.image_u8_to_f32(torch.from_numpy(store).cuda(), torch.from_numpy(ids).cuda()).cpu().numpy()
    ref = (store[ids].astype(np.float32) / 255.0 - 0.5) / 0.5           # ToTensor + Normalize(0.5, 0.5)
    ref = np.transpose(ref, (0, 1, 4, 2, 3))
    ref[ids == 0] = 0.0                                                 # pad image = zeros (trainset.py:96)
    assert out.shape == (2, 3, 3, 16, 12) and np.abs(out - ref).max() < 1e-6


def test_pixelnet_trainer_end_to_end(tmp_path):
    """main.py surface for the pixel model: YAML -> data -> MOSASRec -> Trainer.fit/evaluate with the HBM-resident
    image store (synthetic covers), two optimizer groups, full-sort eval over encoder-produced item features."""
    import os

    from pixelrec_amd.config import Config
    from pixelrec_amd.data import bulid_dataloader, load_data
    from pixelrec_amd.data.images import interleave_pos_neg
    from pixelrec_amd.optim import OptimizerGroup
    from pixelrec_amd.parallel import DataParallel
    from pixelrec_amd.trainer import Trainer
    from pixelrec_amd.utils import get_model
    from tests.golden_util import GOLDEN_DIR

    my, ov = tmp_path / "m.yaml", tmp_path / "o.yaml"
    my.write_text("model: MOSASRec\nn_layers: 2\nn_heads: 2\nembedding_size: 32\ninner_size: 2\nhidden_dropout_prob: 0.1\n"
                  "attn_dropout_prob: 0.1\nhidden_act: 'gelu'\nlayer_norm_eps: 1e-12\ninitializer_range: 0.02\n")
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
    trainer = Trainer(config, DataParallel(model))
    assert isinstance(trainer.optimizer, OptimizerGroup)
    # the on-device batch assembly reproduces MOSEQTrainDataset's order pos_0, neg_0, pos_1, neg_1, ...
    items = torch.tensor([[[0, 3, 5], [0, 0, 9]]])
    assert interleave_pos_neg(items).tolist() == [[0, 0, 3, 0, 5, 9]]
    best, res = trainer.fit(train, valid, saved=True)
    losses = [trainer.train_loss_dict[e] for e in sorted(trainer.train_loss_dict)]
    assert len(losses) == 3 and losses[-1] < losses[0]
    out = trainer.evaluate(test, load_best_model=True)
    assert set(out) == {"recall@5", "recall@10", "ndcg@5", "ndcg@10"}
    assert trainer.item_feature.shape == (dataload.item_num, 32)
    ck = torch.load(trainer.saved_model_file, map_location="cpu", weights_only=False)
    assert any(k.startswith("visual_encoder.item_encoder.vision_model.encoder.layers.2.") for k in ck["state_dict"])


def test_pixelnet_step_replayed_from_a_hip_graph_equals_the_eager_steps(monkeypatch):
    # (bit identity of the capture / replay mechanism: both runs on per-step exact h2 scales.  A captured step runs its gradient
    # planes under the recent steps' scales by default -- round 6, seqcore "stale scales" -- which an eager loop that has not asked for
    # them does not: tests/test_gpu_h2_stale.py holds that mode to the golden bars)
    monkeypatch.setenv("PXR_SEQ_H2_STALE", "0")
    _pixelnet_graph_equals_eager()


def _pixelnet_graph_equals_eager():
    """The whole PixelNet step (tower forward, backward of its trainable block, sequence block, BOTH optimizer groups: reference
    trainer.py:74-96,116-125) captured once and replayed == the eager sequence step for step, dropout on: VisualAdamW's step
    number lives on the device since round 5, like PxrAdamW's."""
    from pixelrec_amd.graph import GraphedTrainStep
    from pixelrec_amd.optim import OptimizerGroup, PxrAdamW, VisualAdamW

    g = torch.Generator().manual_seed(9)
    batches = [(torch.randn(B, 2 * (L + 1), 3, 64, 64, generator=g).cuda(), torch.ones(B, L, dtype=torch.int64).cuda()) for _ in range(7)]
    one = torch.ones((), device="cuda")

    def run(graph):
        m, _, _ = _build()
        m = m.cuda().train()
        m.hidden_dropout_prob = m.attn_dropout_prob = 0.1
        opt = OptimizerGroup(VisualAdamW(m.visual_encoder, lr=1e-3, weight_decay=0.01), PxrAdamW(m, lr=1e-3, weight_decay=0.1))
        losses, gs = [], None
        for i, (img, mk) in enumerate(batches):
            if graph and i >= 2:
                if gs is None:
                    gs = GraphedTrainStep(m, opt, img, mk, warmup=1)     # (its one eager warm-up step trains on batch 2 as well:
                    continue                                              #  the eager run below does the same)
                losses.append(float(gs(img, mk)))
            else:
                opt.zero_grad()
                loss = m((img, mk))
                loss.backward(one)
                opt.step()
                losses.append(float(loss.detach()))
        return losses, {k: v.detach().clone() for k, v in m.state_dict().items()}, [o.step_count for o in opt.opts]

    le, se, ne = run(False)
    lg, sg, ng = run(True)
    assert ne == ng == [len(batches)] * 2
    assert le[:2] == lg[:2] and le[3:] == lg[2:]            # (the graph run's batch-2 loss is inside the capture's warm-up step)
    for k in se:
        assert torch.equal(se[k], sg[k]), k
