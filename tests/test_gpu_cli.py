"""The launcher as a user runs it (reference code/main.py:9-31 -> run.py:18-82): `python main.py --device 0 --config_file <model yaml>
<overall yaml>` as a SUBPROCESS on a small synthetic interaction CSV -- YAML parsing, data loading, the captured training step, validation
after every epoch, the checkpoint of the best epoch, the test evaluation from that checkpoint, exit code 0."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_main_py_trains_validates_checkpoints_and_tests(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_dataset

    synth_dataset.main(str(tmp_path / "data"), 3000, 800)
    (tmp_path / "m.yaml").write_text("model: SASRec\nn_layers: 2\nn_heads: 2\nembedding_size: 64\ninner_size: 2\n"
                                     "hidden_dropout_prob: 0.1\nattn_dropout_prob: 0.1\nhidden_act: 'gelu'\nlayer_norm_eps: 1e-12\n"
                                     "initializer_range: 0.02\n")
    (tmp_path / "o.yaml").write_text(f"seed: 2020\nstate: INFO\nuse_modality: False\nreproducibility: True\n"
                                     f"checkpoint_dir: '{tmp_path}/saved'\nlog_path: '{tmp_path}/log'\nshow_progress: False\n"
                                     f"MAX_ITEM_LIST_LENGTH: 10\ndata_path: {tmp_path}/data/\ndataset: Pixel200K\nepochs: 3\n"
                                     "train_batch_size: 64\noptim_args: {learning_rate: 0.001, weight_decay: 0.1}\n"
                                     "eval_batch_size: 512\ntopk: [5,10]\nmetrics: ['Recall', 'NDCG']\nvalid_metric: NDCG@10\n"
                                     "metric_decimal_place: 7\neval_step: 1\nstopping_step: 30\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "OMP_NUM_THREADS")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--device", "0", "--config_file", str(tmp_path / "m.yaml"),
                        str(tmp_path / "o.yaml")], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert len(re.findall(r"epoch \d+ training \[time", out)) == 3 and len(re.findall(r"epoch \d+ evaluating \[time", out)) == 3, out[-3000:]
    m = re.search(r"test result: .*?'ndcg@10', ([0-9.]+)\)", out)
    assert m is not None and 0.0 <= float(m.group(1)) <= 1.0, out[-2000:]
    saved = [f for f in os.listdir(tmp_path / "saved") if f.endswith(".pth")]
    assert len(saved) == 1
