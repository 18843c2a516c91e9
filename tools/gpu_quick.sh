#!/bin/bash
# Scratch runner for one-off GPU experiments:  gpurun --timeout N -- 'bash tools/gpu_quick.sh'
# Edit the body for the experiment at hand (results under gpurun_out/, copy what should be judged into profiles/).  The stable entry points are
# tools/gpu_check.sh (tests / smoke / bench / profiles by stage name) and tools/run_profiles.sh (the full profile set).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision"
( for i in 1 2; do
    echo "## default library"; timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], {k:v['ms'] for k,v in d['kernels_ms'].items() if k.startswith('voc') or k.startswith('dec')})"
    echo "## derived remainder scale (EV_MXQ_FIXED_LO)"; EVHIP_LIB=emotivoice_amd/csrc/libevhip_fixlo.so timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], {k:v['ms'] for k,v in d['kernels_ms'].items() if k.startswith('voc') or k.startswith('dec')})"
  done ) > gpurun_out/fixlo_ab.txt 2>&1; echo "rc=$?"; cat gpurun_out/fixlo_ab.txt | cut -c1-400
