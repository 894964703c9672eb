timeout 300 python profiles/sort_times.py c1 2>&1 | grep points
timeout 300 python profiles/sort_times.py c2 2>&1 | grep points
