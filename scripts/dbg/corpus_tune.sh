#!/bin/bash
# tryout: convergence of clstmocrtrain on the rendered corpus; CFGS="lr,updates,batch ..."
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/corpus_tune"; mkdir -p "$OUT"
make -C clstm_amd/host -s all
python scripts/make_corpus.py /tmp/corpus --n 512 --faces ${FACES:-6} ${COPTS} > "$OUT/corpus.txt"
head -64 /tmp/corpus/list.txt > /tmp/corpus/test.txt
for cfg in ${CFGS:-"1e-4,1000,64"}; do
  IFS=, read lr up bt <<< "$cfg"
  ( time batch=$bt ntrain=$((bt*up)) lrate=$lr nhidden=100 seed=0.222 save_name=/tmp/corpus/_m save_every=100000000 report_every=$((bt*up/5)) \
    test_every=$((bt*up/5)) clstm_amd/bin/clstmocrtrain /tmp/corpus/list.txt /tmp/corpus/test.txt ) > "$OUT/train_${lr}_${up}_${bt}.log" 2>&1
  echo "== lr $lr updates $up batch $bt"; grep -E "^ERROR|^OUT|^TRU|real" "$OUT/train_${lr}_${up}_${bt}.log" | tail -9
done
