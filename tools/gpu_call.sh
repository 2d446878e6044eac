#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/prof_jk_host.py --world 8 2>&1 | tail -45
