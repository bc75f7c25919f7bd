# the contention test at the current library, and — to see that the detector detects — at a library built from the same sources WITHOUT the barrier of be_accept_body
cd $GRAFT_REPO_ROOT
echo "== current library"; python -m pytest tests/test_contention.py -q -m gpu 2>&1 | grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | tail -4
rm -rf /tmp/old /tmp/oldlib && mkdir -p /tmp/old/dynamic_vins_amd /tmp/oldlib && cp -r dynamic_vins_amd/csrc /tmp/old/dynamic_vins_amd/csrc && cp -r include /tmp/old/include && rm -rf /tmp/old/dynamic_vins_amd/csrc/build
python - <<'PY'
p='/tmp/old/dynamic_vins_amd/csrc/be_kernels.h'; s=open(p).read()
a="    __syncthreads();\n    if (c.done || !c.pending) return;          // failed factorisations"
assert s.count(a)==1
open(p,'w').write(s.replace(a,"    if (c.done || !c.pending) return;          // failed factorisations"))
PY
(cd /tmp/old/dynamic_vins_amd/csrc && make -j 16 OUT=/tmp/oldlib/libdvins_hip.so > /tmp/oldbuild.log 2>&1; tail -2 /tmp/oldbuild.log | cut -c1-200; ls -la /tmp/oldlib/)
for k in 1 2 3; do echo "== PRE-FIX library, pass $k"; DVINS_HIP_LIB=/tmp/oldlib/libdvins_hip.so python -m pytest tests/test_contention.py -q -m gpu 2>&1 | grep -v "RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" | grep "passed\|failed\|differs" | cut -c1-220 | tail -4; done
