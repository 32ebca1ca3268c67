#!/bin/bash
bash tools/c2_quick.sh
for V in physicsbasedfwi2_amd/_variants/*.so; do echo "== $V"; MIFWI_LIB=$GRAFT_REPO_ROOT/$V bash tools/c2_quick.sh; done
