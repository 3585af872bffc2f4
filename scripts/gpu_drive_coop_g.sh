app=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(gpu=True))")
for g in 0 2 3 4 6 8; do
  if [ $g = 0 ]; then unset KBA_COOP_G; else export KBA_COOP_G=$g; fi
  echo "== KBA_COOP_G=$g"; timeout 300 $app --frames 600 --az 2000 --quiet 2>&1 | grep -E "^limo_stream: (pipeline|host)" | sed 's/input synthesis.*//'
done
