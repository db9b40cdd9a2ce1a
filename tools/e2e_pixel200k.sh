#!/bin/bash
# End-to-end run of the main.py surface on a Pixel200K-shaped synthetic dataset (BASELINE configs[0] shape:
# SASRec IDNet emb=128 seq_len=20) -- data pipeline + training epochs + full-sort evaluation, on one MI355X.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/e2e; rm -rf $W; mkdir -p $W/data $W/cfg
cd $REPO
python tools/synth_dataset.py $W/data ${N_USERS:-200000} ${N_ITEMS:-96000}
cat > $W/cfg/model.yaml <<Y
model: SASRec
n_layers: 2
n_heads: 4
embedding_size: ${EMB:-128}
inner_size: 2
hidden_dropout_prob: 0.1
attn_dropout_prob: 0.1
hidden_act: 'gelu'
layer_norm_eps: 1e-12
initializer_range: 0.02
${EXTRA_MODEL_YAML:-}
Y
cat > $W/cfg/overall.yaml <<Y
seed: 2020
state: INFO
use_modality: False
reproducibility: True
checkpoint_dir: '$W/saved'
log_path: '$W/log'
show_progress: False
MAX_ITEM_LIST_LENGTH: ${SEQ:-20}
data_path: $W/data/
dataset: Pixel200K
epochs: ${EPOCHS:-3}
train_batch_size: 64
optim_args: {learning_rate: 0.0001, weight_decay: 0.1}
eval_batch_size: 1024
topk: [5,10]
metrics: ['Recall', 'NDCG']
valid_metric: NDCG@10
metric_decimal_place: 7
eval_step: 1
stopping_step: 30
Y
cd $W && time python $REPO/main.py --device 0 --config_file $W/cfg/model.yaml $W/cfg/overall.yaml 2>&1 | grep -v "amdgpu.ids" | tail -25
