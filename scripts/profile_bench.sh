#!/bin/bash
# The headline configuration's rocprofv3 evidence (kernel trace + stats, FETCH_SIZE / WRITE_SIZE, SQ passes) and its entry of
# profiles/traffic.json: since round 6 one configuration among five of scripts/profile_configs.sh.
# usage: scripts/profile_bench.sh <tag>
exec "$(dirname "$0")/profile_configs.sh" "$1" M
