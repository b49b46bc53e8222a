"""Set up paths (same role as the reference's tools/_init_paths.py:1-15)."""
import os.path as osp
import sys

this_dir = osp.dirname(osp.abspath(__file__))
for p in (osp.join(this_dir, '..'), osp.join(this_dir, '..', 'lib')):
    p = osp.abspath(p)
    if p not in sys.path:
        sys.path.insert(0, p)
