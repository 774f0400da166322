# the round's last run of tools/r05_profile.sh: everything GPU-side (tests, PMC, bench lines of every workload, traces, long run); the host-side legs
# (feed, CLI traces, database load, stress checks) did not change since their last run and are skipped
R05_SKIP_HOST=1 bash tools/r05_profile.sh
