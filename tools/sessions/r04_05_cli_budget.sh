cd $GRAFT_REPO_ROOT
python - <<'PY'
import importlib, numpy as np, os
synth = importlib.import_module("optical-flow-1_amd.synth")
I0, I1 = synth.pair("P1", 1920, 1080)
for name, img in (("a.pgm", I0), ("b.pgm", I1)):
    with open("/dev/shm/" + name, "wb") as f:
        f.write(b"P5\n1920 1080\n255\n"); f.write(np.clip(img, 0, 255).astype(np.uint8).tobytes())
PY
for i in 1 2 3; do
  time env OFX_TRACE_INIT=1 OFX_STATS=- optical-flow-1_amd/bin/tvl1flow /dev/shm/a.pgm /dev/shm/b.pgm /dev/shm/out.flo 0 0.25 0.15 0.3 5 0.5 5 0.01 0 2>&1 | cut -c1-400
done
echo "== LD_DEBUG statistics"
LD_DEBUG=statistics optical-flow-1_amd/bin/tvl1flow /dev/shm/a.pgm /dev/shm/b.pgm /dev/shm/out.flo 0 0.25 0.15 0.3 5 0.5 5 0.01 0 2>&1 | grep -E "total startup|relocation|load" | head
ls -la optical-flow-1_amd/libofx.so; ldd optical-flow-1_amd/bin/tvl1flow | head -20
rm -f /dev/shm/a.pgm /dev/shm/b.pgm /dev/shm/out.flo
