for cfg in "X=1" "BP_NOSTEM=1" "BP_NOTINY=1" "BP_EPILOGUE_STATS=0" "BP_IGEMM_PPCIN=1000000"; do
  echo "== $cfg"
  env $cfg python tools/grad_report.py fid128_n2 128 2 mfma 2>&1 | grep -v amdgpu | head -6
done
