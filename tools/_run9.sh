cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "vae or conv3d" 2>&1 | tail -4
python tools/vae_trace.py both fp32 2>&1 | tail -2
