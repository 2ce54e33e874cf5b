timeout 1800 python -X faulthandler -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests_c19.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Thread\|^$" gpurun_out/gpu_tests_c19.log | head -30 | cut -c1-300; tail -3 gpurun_out/gpu_tests_c19.log
python -c "import __graft_entry__ as g; g.smoke()"
bash scripts/r2_profile.sh > gpurun_out/r2prof.log 2>&1; tail -3 gpurun_out/r2prof.log
