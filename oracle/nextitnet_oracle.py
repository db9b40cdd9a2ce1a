"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's NextItNet (code/REC/model/IDNet/nextitnet.py:58-113 and the
residual block (b) of :160-194), the causal dilated convolution written as explicit shifted sums (no nn.Conv2d).  Imported by
tests/ only.  Pinned against the reference itself: tests/golden/nextitnet_tiny.npz is written by
oracle/make_golden_nextitnet.py from `REC.model.IDNet.nextitnet.NextItNet` run unmodified; tests/test_nextitnet_golden.py
checks this file against it.
"""
import torch


def causal_conv(x, w, b, dilation):
    """x [B, L, C_in], w [C_out, C_in, 1, k] (the Conv2d parameter), left zero padding of (k-1) dilation (nextitnet.py:183-194)."""
    B, L, _ = x.shape
    k = w.shape[-1]
    out = b.expand(B, L, -1).clone()
    for j in range(k):
        shift = (k - 1 - j) * dilation
        if shift >= L:
            continue
        xs = torch.zeros_like(x)
        xs[:, shift:] = x[:, :L - shift]
        out = out + xs @ w[:, :, 0, j].t()
    return out


def layer_norm(x, w, b, eps=1e-8):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def encode(params, seq_ids, dilations):
    return encode_rows(params, params["item_embedding.weight"][seq_ids], dilations)


def encode_rows(params, x, dilations):
    for i, d in enumerate(dilations):
        p = lambda n: params[f"residual_blocks.{i}.{n}"]
        o = torch.relu(layer_norm(causal_conv(x, p("conv1.weight"), p("conv1.bias"), d), p("ln1.weight"), p("ln1.bias")))
        o = torch.relu(layer_norm(causal_conv(o, p("conv2.weight"), p("conv2.bias"), 2 * d), p("ln2.weight"), p("ln2.bias")))
        x = o + x
    if "final_layer.weight" in params:
        x = x @ params["final_layer.weight"].t() + params["final_layer.bias"]
    return x


def forward_loss(params, items, masked_index, dilations):
    """nextitnet.py:58-78."""
    pos_ids, neg_ids = items[:, 0], items[:, 1]
    out = encode(params, pos_ids[:, :-1], dilations)
    E = params["item_embedding.weight"]
    pos = (out * E[pos_ids[:, 1:]]).sum(-1)
    neg = (out * E[neg_ids[:, 1:]]).sum(-1)
    loss = -(torch.log((pos - neg).sigmoid() + 1e-8) * masked_index).sum(-1)
    return loss.mean(-1)


@torch.no_grad()
def predict(params, item_seq, item_feature, dilations):
    """nextitnet.py:92-106 (the sequence is embedded with the model's own table)."""
    out = encode(params, item_seq, dilations)
    return out[:, -1] @ item_feature.t()


def forward_loss_rows(params, item_emb, masked_index, dilations):
    """PixelNet's MONextItNet.forward after the image encoder (code/REC/model/PixelNet/monextitnet.py:55-73): item_emb
    [B, L+1, 2, D] with pos | neg interleaved."""
    pos, neg = item_emb[:, :, 0], item_emb[:, :, 1]
    out = encode_rows(params, pos[:, :-1], dilations)
    ps, ns = (out * pos[:, 1:]).sum(-1), (out * neg[:, 1:]).sum(-1)
    return (-(torch.log((ps - ns).sigmoid() + 1e-8) * masked_index).sum(-1)).mean(-1)
