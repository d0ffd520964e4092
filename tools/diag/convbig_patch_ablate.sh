for fl in "" "-DPV_PATCH_ABLATE=1" "-DPV_PATCH_ABLATE=2" "-DPV_PATCH_ABLATE=3" "-DPV_PATCH_ABLATE=4"; do
  echo "=== flags: $fl"
  CB_FLAGS="$fl" timeout 300 python tools/diag/convbig_seg_stamps.py 2>&1 | grep -A3 "PV_CONV_PATCH=1" | grep -v "^--"
done
