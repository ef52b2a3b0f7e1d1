#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -6
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep smoke
