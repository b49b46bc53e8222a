cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 300 --tb=line 2>&1 | cut -c1-200 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(cd tf-faster-rcnn_amd/tools && timeout 200 python test_net.py --net res50 --imdb synthetic_3 2>&1 | tail -2; timeout 200 python trainval_net.py --net res50 --iters 4 --set TRAIN.DISPLAY 2 TRAIN.BG_THRESH_LO 0.0 2>&1 | tail -4)
