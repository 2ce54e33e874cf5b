for r in 16 64 128 192 224; do PBSGPU_SCAN_CU_RESERVE=$r python scripts/r2_probe_clock.py 2>&1 | tail -2; done
