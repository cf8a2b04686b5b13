#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for ns in 3 4 5; do for live in 700 917 1100 1400; do echo "NS=$ns live=$live: $(AZK_TAIL_NS=$ns timeout -k 10 100 python3 tools/run_tail2.py 2048 $live 100 2>&1 | grep 'round 1 lds=True')"; done; done
