cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/diag/refresh_profiles.sh 2>&1 | tail -45 | cut -c1-400
