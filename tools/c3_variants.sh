#!/bin/bash
# C3 timing of the in-tree library and of every variant under physicsbasedfwi2_amd/_variants (MIFWI_LIB)
bash tools/c3_quick.sh
for V in physicsbasedfwi2_amd/_variants/*.so; do echo "== $V"; MIFWI_LIB=$GRAFT_REPO_ROOT/$V bash tools/c3_quick.sh; done
