#!/bin/bash
# Scratch wrapper for one `gpurun` call (overwritten per experiment).  The canonical commands are:
#   python -m pytest tests -m gpu -q          python bench.py          bash tools/profile_round.sh r02
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -q -x
