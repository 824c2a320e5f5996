#!/bin/bash
# round 6, call r: end to end on the DEFAULT path: the four models through Trainer.train (epochs + evaluation + test) on the yelp-shaped
# synthetic graph, LightGCN also at amazon-book size
O=gpurun_out/r06r; mkdir -p $O
timeout 900 python tools/e2e_defaults.py yelp 2 > $O/e2e_yelp.log 2>&1; echo "e2e yelp rc $?"; tail -1 $O/e2e_yelp.log | cut -c1-1500
timeout 900 python tools/e2e_defaults.py amazon-book 2 lightgcn,simgcl > $O/e2e_amazon.log 2>&1; echo "e2e amazon rc $?"; tail -1 $O/e2e_amazon.log | cut -c1-900
