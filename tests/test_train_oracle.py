"""The training-step oracle against the fixtures written from the reference's AdapterModel
(oracle/gen_golden_train.py; train.py:258-296)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import train_oracle as TO

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "train_*.npz")))


def load_case(path):
    z = np.load(path)
    d, H, Hkv, V, steps, warm = (int(x) for x in z["meta"])
    params = {k: torch.from_numpy(z[f"init.{k}"]).clone() for k in TO.PARAM_NAMES}
    batches = [{k: torch.from_numpy(z[f"batch{i}.{k}"]) for k in ("hidden_states_early", "target", "attention_mask", "loss_mask")} for i in range(steps)]
    return z, (d, H, Hkv, V, steps, warm), params, batches


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_training_steps_match_the_reference(path):
    z, (d, H, Hkv, V, steps, warm), params, batches = load_case(path)
    head = torch.from_numpy(z["head"])
    opt = TO.AdamWState(params)
    base_lr = float(z["base_lr"][0])
    for i, batch in enumerate(batches):
        lr = TO.linear_warmup_lr(base_lr, i, warm, 100)
        assert abs(lr - float(z["lr"][i])) < 1e-12
        if i == 0:
            predict = TO.adapter_forward(batch["hidden_states_early"], batch["attention_mask"], params, H, Hkv, 1e-6)
            assert torch.allclose(predict, torch.from_numpy(z["predict0"]), atol=2e-5, rtol=1e-5)
        loss, acc, grads = TO.train_step(params, opt, head, batch, H, Hkv, 1e-6, lr)
        assert abs(loss - float(z["loss"][i])) < 2e-5 * max(1.0, abs(loss))
        assert abs(acc - float(z["prob_acc"][i])) < 2e-6
        if i == 0:
            for k in TO.PARAM_NAMES:
                g = torch.from_numpy(z[f"grad0.{k}"])
                assert torch.allclose(grads[k], g, atol=1e-6 + 1e-4 * float(g.abs().max())), k
    for k in TO.PARAM_NAMES:
        assert torch.allclose(params[k], torch.from_numpy(z[f"final.{k}"]), atol=2e-5, rtol=1e-5), k


def test_the_first_step_runs_at_zero_learning_rate():
    # get_linear_schedule_with_warmup starts at factor 0 (train.py:235): the first optimizer step changes nothing
    assert TO.linear_warmup_lr(5e-5, 0, 2000, 800000) == 0.0
    assert TO.linear_warmup_lr(5e-5, 2000, 2000, 800000) == 5e-5
