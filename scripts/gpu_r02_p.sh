#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash scripts/prof_chain.sh p 32 2>&1 | grep -v simple_timer | head -30
