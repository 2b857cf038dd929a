for d in 0 1 2 3 7; do echo "RR_DEBUG=$d"; RR_DEBUG=$d python scripts_probe.py 2 1 2 200 300 2>&1 | grep -E "column|frames"; done
for d in 0 1 2 3; do echo "nonoise RR_DEBUG=$d"; RR_DEBUG=$d python scripts_probe.py 2 1 0 200 300 2>&1 | grep -E "column"; done
