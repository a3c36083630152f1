"""Pin the CPU restatement (oracle/) to vectors produced by the reference's own code.

The .npz files under tests/golden were written by oracle/gen_golden.py, which runs
the reference's unmodified kangaroo_forward / AdapterModel / layer classes."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import kangaroo_oracle as ko

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLD, "*.npz")) if not p.endswith("layers.npz") and not os.path.basename(p).startswith("train_"))


def load_case(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    base = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("base/")}
    adapter = {k[8:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("adapter/")}
    return z, meta, base, adapter


def oracle_shape(name):
    from kangaroo_amd.config import get_shape
    s = get_shape(name)
    return ko.Shape(**s.to_dict())


def test_fixtures_present():
    assert len(CASES) >= 6
    assert os.path.exists(os.path.join(GOLD, "layers.npz"))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_loop_matches_reference(path):
    z, meta, base, adapter = load_case(path)
    shape = oracle_shape(meta["shape"])
    for pi in range(meta["n_prompts"]):
        model = ko.OracleKangarooModel(shape, base, adapter, meta["exit_layer"])
        ids = torch.from_numpy(z[f"p{pi}/input_ids"])
        tr = ko.Trace()
        out, new_token, idx, accept = ko.kangaroo_forward(
            ids, model, meta["eos"], meta["max_new_tokens"], EARLY_STOP_LAYER=meta["exit_layer"],
            SPECULATIVE_DECODING_STEPS=meta["steps"], threshold=meta["threshold"], trace=tr, keep_logits=True,
            accept_top_k=meta.get("accept_top_k", 1), accept_top_p=meta.get("accept_top_p"),
            temperature=meta.get("temperature") or 1.0, accept_typical=meta.get("accept_typical"))
        assert out[0] == z[f"p{pi}/output_ids"].tolist()
        assert new_token == int(z[f"p{pi}/new_token"])
        assert idx == int(z[f"p{pi}/idx"])
        assert accept == z[f"p{pi}/accept_length_list"].tolist()
        assert tr.draft_lengths == z[f"p{pi}/draft_lengths"].tolist()
        np.testing.assert_allclose(np.asarray(tr.p_max), z[f"p{pi}/p_max"], rtol=0, atol=1e-5)
        if tr.draft_logits:
            np.testing.assert_allclose(torch.stack(tr.draft_logits).numpy(), z[f"p{pi}/draft_logits"], rtol=0, atol=1e-5 * max(1.0, float(np.abs(z[f"p{pi}/draft_logits"]).max())))
        vl = torch.cat(tr.verify_logits, 0).numpy()
        np.testing.assert_allclose(vl, z[f"p{pi}/verify_logits"], rtol=0, atol=1e-5 * max(1.0, float(np.abs(vl).max())))


@pytest.mark.parametrize("path", CASES[:4], ids=[os.path.basename(p)[:-4] for p in CASES[:4]])
def test_lossless_vs_vanilla(path):
    """Kangaroo's greedy output equals plain greedy decoding of the same model."""
    z, meta, base, adapter = load_case(path)
    shape = oracle_shape(meta["shape"])
    for pi in range(meta["n_prompts"]):
        model = ko.OracleKangarooModel(shape, base, adapter, meta["exit_layer"])
        ids = torch.from_numpy(z[f"p{pi}/input_ids"])
        gold = z[f"p{pi}/output_ids"].tolist()
        van, _ = ko.vanilla_forward(ids, model, meta["eos"], len(gold) - ids.shape[1])
        assert van == gold


def test_layer_vectors():
    z = np.load(os.path.join(GOLD, "layers.npz"))
    t = lambda k: torch.from_numpy(z[k])
    y = ko.rmsnorm(t("rms/x")[0], t("rms/w"), 1e-6)
    np.testing.assert_allclose(y.numpy(), z["rms/y"][0], atol=1e-6)
    cos, sin = ko.rope_tables(16, 2048)
    np.testing.assert_allclose(cos[:40].numpy(), z["rope/cos"], atol=1e-6)
    np.testing.assert_allclose(sin[:40].numpy(), z["rope/sin"], atol=1e-6)
    qe, ke = ko.apply_rope(t("rope/q")[0], t("rope/k")[0], cos, sin, t("rope/pos")[0])
    np.testing.assert_allclose(qe.numpy(), z["rope/qe"][0], atol=1e-6)
    np.testing.assert_allclose(ke.numpy(), z["rope/ke"][0], atol=1e-6)
    y = ko.mlp(t("mlp/x")[0], t("mlp/wg"), t("mlp/wu"), t("mlp/wd"))
    np.testing.assert_allclose(y.numpy(), z["mlp/y"][0], atol=1e-5)
    shape = ko.Shape(64, 1, 4, 2, 128, 97)
    aw = {k[8:]: t(k) for k in z.files if k.startswith("adapter/")}
    ad = ko.OracleAdapter(shape, aw)
    y0, kv = ad.forward_early_stop(inputs_embeds=t("ad/x0"), position_ids=torch.arange(6).view(1, -1))
    np.testing.assert_allclose(y0.numpy(), z["ad/y0"], atol=2e-5)
    y1, kv = ad.forward_early_stop(inputs_embeds=t("ad/x1"), position_ids=torch.tensor([[6]]), past_key_values=kv)
    np.testing.assert_allclose(y1.numpy(), z["ad/y1"], atol=2e-5)
    y2, kv = ad.forward_early_stop(inputs_embeds=t("ad/x2"), position_ids=torch.tensor([[7, 8]]), past_key_values=kv)
    np.testing.assert_allclose(y2.numpy(), z["ad/y2"], atol=2e-5)
    np.testing.assert_allclose(kv[0][0].numpy(), z["ad/k2"], atol=2e-5)
    np.testing.assert_allclose(kv[0][1].numpy(), z["ad/v2"], atol=2e-5)


def test_error_conventions():
    """kangaroo/earlyexit.py:16-25."""
    z, meta, base, adapter = load_case(CASES[0])
    model = ko.OracleKangarooModel(oracle_shape(meta["shape"]), base, adapter, 2)
    with pytest.raises(AssertionError):
        model.base_model.forward_draft_or_large_model(in_tokens_small=torch.tensor([[3]]))
    model.base_model.past_key_values = []
    with pytest.raises(ValueError):
        model.base_model.forward_draft_or_large_model()
    with pytest.raises(ValueError):
        model.base_model.forward_draft_or_large_model(in_tokens_small=torch.tensor([[3]]), in_features_large=torch.zeros(1, 1, 64))
