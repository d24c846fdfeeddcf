#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import _oracle as O
text,_=O.corpus(24, seed=100, variants=O.VARIANTS_C2, plant_period=30)
open('/tmp/f0.txt','wb').write(text.tobytes())
PY
for a in "-V0 -2" "--gpus 1 -V0 -2" "--gpus 1 -V0 -2 -c" "--gpus 1 -V0 -n -i -2"; do
  ./agrep_amd/agrep-hip $a approximatematch /tmp/f0.txt > /tmp/o.txt 2>/tmp/e.txt; echo "[$a] rc=$? out=$(wc -c < /tmp/o.txt) err=$(head -c 300 /tmp/e.txt)"
done
AGH_DEBUG=1 ./agrep_amd/agrep-hip --gpus 1 -V0 -2 approximatematch /tmp/f0.txt 2>&1 | tail -5
