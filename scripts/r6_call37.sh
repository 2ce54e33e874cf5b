#!/bin/bash
# round 6, call 37: the single-wave lanes form with TWO waves per SIMD (eight single-wave workgroups per CU)
out=gpurun_out/r6c37; mkdir -p $out
timeout 500 python scripts/r6_sha_forms.py 1048576 131072 > $out/forms_1m_131072.log 2>&1; cat $out/forms_1m_131072.log | tail -6
timeout 500 python scripts/r6_sha_forms.py 2097152 65536 > $out/forms_2m_65536.log 2>&1; cat $out/forms_2m_65536.log | tail -6
