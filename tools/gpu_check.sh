#!/bin/bash
# Run on the GPU box via:  gpurun --timeout 2400 -- 'bash tools/gpu_check.sh [stage...]'
# Stages: ops parity gen smoke bench ubench prof pmc pmccal pmcsq dist2    (default: ops parity smoke bench)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
STAGES="${@:-ops parity smoke bench}"
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.log
nproc >> gpurun_out/device.log
python -c "import bench; print(bench.csrc_hash())" > gpurun_out/csrc_hash.txt      # what every measurement of this call ran on (profile_summary.py stamps it)
for st in $STAGES; do
  case $st in
    ops)    timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 > gpurun_out/ops.log 2>&1; echo "ops rc=$?";;
    parity) timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 > gpurun_out/parity.log 2>&1; echo "parity rc=$?";;
    smoke)  timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?";;
    bench)  timeout 900 python bench.py --steps 10 --warmup 3 --cpu-utts 4 --dump-launches gpurun_out/launches.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?";;
    gen)    timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_api.py tests/test_gpu_simbert.py tests/test_gpu_dist.py -m gpu -q --timeout 600 > gpurun_out/gen.log 2>&1; echo "gen rc=$?";;
    pmc)    cd /tmp && export TMPDIR=/tmp
            timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_fetch" -o f -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --cpu-utts 0 --no-other-precision > "$GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log" 2>&1
            timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_write" -o w -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --cpu-utts 0 --no-other-precision > "$GRAFT_REPO_ROOT/gpurun_out/pmc_write.log" 2>&1
            echo "pmc rc=$?"; cd "$GRAFT_REPO_ROOT";;
    pmcsq)  # MFMA-busy / wait / LDS-conflict counters (separate passes, --kernel-trace only; never combined with sys/hip/hsa traces)
            cd /tmp && export TMPDIR=/tmp
            rocprofv3 -L 2>/dev/null | grep -E "SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CU_CYCLES|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_LDS_UNALIGNED_STALL|GRBM_GUI_ACTIVE|SQ_INSTS_VALU_MFMA|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY" | head -40 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_available.txt"
            BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-other-precision"
            timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_sq" -o s -- $BENCH > "$GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log" 2>&1; echo "pmc_sq rc=$?"
            timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_lds" -o l -- $BENCH > "$GRAFT_REPO_ROOT/gpurun_out/pmc_lds.log" 2>&1; echo "pmc_lds rc=$?"
            cd "$GRAFT_REPO_ROOT";;
    pmccal) # calibration of FETCH_SIZE / WRITE_SIZE on a copy of known size (MI355X_MICROARCH.md section HBM)
            cd /tmp && export TMPDIR=/tmp
            timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_cal_fetch" -o c -- python "$GRAFT_REPO_ROOT/tools/pmc_calibrate.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_cal_fetch.log" 2>&1
            timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_cal_write" -o c -- python "$GRAFT_REPO_ROOT/tools/pmc_calibrate.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_cal_write.log" 2>&1
            echo "pmccal rc=$?"; cd "$GRAFT_REPO_ROOT";;
    dist2)  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --force-device 0 --batch 8 > gpurun_out/dist2.log 2>&1; echo "dist2 rc=$?"; tail -n 3 gpurun_out/dist2.log | cut -c1-600;;
    ubench) # stand-alone micro-benchmarks / hardware checks built in-tree (hipcc -o tools/build/<name> tools/<name>.hip): layout A/B of the plane sets,
            # the K32 -> K16 MFMA chain check; each prints its own verdict line
            for exe in tools/build/*; do [ -x "$exe" ] && timeout 300 "$exe" > "gpurun_out/ubench_$(basename $exe).txt" 2>&1; echo "ubench $(basename $exe) rc=$?"; tail -n 12 "gpurun_out/ubench_$(basename $exe).txt" | cut -c1-400; done;;
    prof)   cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --cpu-utts 0 --no-other-precision --voc-streams 1 > "$GRAFT_REPO_ROOT/gpurun_out/prof.log" 2>&1; echo "prof rc=$?"; cd "$GRAFT_REPO_ROOT";;
  esac
done
tail -n 30 gpurun_out/ops.log gpurun_out/parity.log gpurun_out/gen.log gpurun_out/smoke.log gpurun_out/bench.log 2>/dev/null | tail -n 120
