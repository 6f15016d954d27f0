#!/bin/bash
# end-to-end rate of the C++ driver from PNG files: the fixture 256 times, minibatches of 64 (helper threads read + normalise
# the next minibatch; minibatches between reports are enqueued without host synchronisation)
TAG=${1:-drv}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
make -C clstm_amd/host -s all
W=$(mktemp -d); cp tests/golden/textline.bin.png "$W/l.bin.png"; cp tests/golden/textline.gt.txt "$W/l.gt.txt"
for i in $(seq 256); do echo "$W/l.bin.png"; done > "$W/list.txt"
for B in 64; do
( cd "$W"; TIMEFORMAT="wall %R s"; time env batch=$B ntrain=25600 report_every=6400 report_time=1 save_name="" test_every=1000000 hidden=100 lrate=1e-4 "$ROOT/clstm_amd/bin/clstmocrtrain" list.txt ) > "$OUT/driver_batch$B.log" 2>&1
grep -E "steptime|wall" "$OUT/driver_batch$B.log"
done
python - "$W/l.bin.png" <<'PY'
import sys
from PIL import Image
im = Image.open(sys.argv[1]); print("fixture", im.size)
PY
