# Same-box A/B of the weight-gradient riders (MMDFN_WGRAD_RIDERS) on the bench workloads
bash tools/ab_env.sh MMDFN_WGRAD_RIDERS 0 1 "$@"
