# round 6, run 24: the streamed frame through a map in reference-order mode (the header shim's default), update on the device
mkdir -p gpurun_out/r06
( echo "== default (fast) map"; timeout 600 python profiles/stream_probe.py 2>&1 | tail -4
  echo "== reference-order map (SAGEICP_MAP_REFERENCE_ORDER=1: what the header shim creates), update on the device"
  SAGEICP_MAP_REFERENCE_ORDER=1 timeout 600 python profiles/stream_probe.py 2>&1 | tail -4
  echo "== reference-order map, update on the host (round 5)"
  SAGEICP_MAP_REFERENCE_ORDER=1 STREAM_HOST_MAP_UPDATE=1 timeout 600 python profiles/stream_probe.py 2>&1 | tail -4
  echo "== reference-order map, LocalMap() after every frame"
  SAGEICP_MAP_REFERENCE_ORDER=1 STREAM_LOCALMAP=1 timeout 600 python profiles/stream_probe.py 2>&1 | tail -5 ) | tee gpurun_out/r06/stream_reference_order.txt
