#!/bin/bash
for N in ${NAPS:-1 4 16 48}; do echo "== nap $N"; MIFWI_POLL_NAP=$N bash tools/c2_quick.sh; done
