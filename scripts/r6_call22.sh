#!/bin/bash
# round 6, call 22: per-CU throughput of the SHA-256 kernel forms on full lanes
out=gpurun_out/r6c22; mkdir -p $out
timeout 900 python scripts/r6_sha_forms.py 2097152 65536 > $out/forms_2m_full.log 2>&1; tail -4 $out/forms_2m_full.log
timeout 900 python scripts/r6_sha_forms.py 4194304 32768 > $out/forms_4m_half.log 2>&1; tail -4 $out/forms_4m_half.log
