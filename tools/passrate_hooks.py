"""EXPERIMENT hooks for tools/passrate_probe.py (round 6): patches a working tree with the -DRT_COUNT_NODES code (per-node box-test counters in the
filtered walk, a contraction set read from the file RT_COLLAPSE_SET names), to be built by tools/mkcount.sh and reverted with
`git checkout -- raytracinginrust_amd/csrc` afterwards: none of it is product code."""
import re
import os; root=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracinginrust_amd", "csrc") + "/"
# rt_ir.h: field under ifdef
p=root+'rt_ir.h'; s=open(p).read()
old="    float rect_m;\n};"
new="    float rect_m;\n#ifdef RT_COUNT_NODES      // tools/passrate_probe.py (experiment build only): per BVH node {box tests, passes} of the filtered walk's box steps\n    unsigned long long* node_counts;\n#endif\n};"
assert old in s; s=s.replace(old,new,1); open(p,'w').write(s)
# kernel: lock-step filt box_step + persistent box_step
p=root+'rt_kernel.hip'; s=open(p).read()
old='''                const DFNode nd = fetch_fnode<ALL>(P, node);
                const bool pass = filter_pass(nd, F);
                const bool ahead = SPEC && pass && (nd.info & FNODE_LEAF) != 0u && p1 == ST_DONE;      // first pending leaf: remember it, walk on'''
new='''                const DFNode nd = fetch_fnode<ALL>(P, node);
                const bool pass = filter_pass(nd, F);
#ifdef RT_COUNT_NODES
                if (P.node_counts) { const uint32_t cid = id_of(P, node); atomicAdd(&P.node_counts[2u * cid], 1ull); if (pass) atomicAdd(&P.node_counts[2u * cid + 1u], 1ull); }
#endif
                const bool ahead = SPEC && pass && (nd.info & FNODE_LEAF) != 0u && p1 == ST_DONE;      // first pending leaf: remember it, walk on'''
assert old in s; s=s.replace(old,new,1)
old='''                        auto box_step = [&]() { const DFNode nd = fetch_fnode<ALL>(P, tv_node); tv_node = filter_pass(nd, F) ? nd.info : nd.skip; };'''
new='''#ifdef RT_COUNT_NODES
                        auto box_step = [&]() { const DFNode nd = fetch_fnode<ALL>(P, tv_node); const bool pass = filter_pass(nd, F);
                                                if (P.node_counts) { const uint32_t cid = id_of(P, tv_node); atomicAdd(&P.node_counts[2u * cid], 1ull); if (pass) atomicAdd(&P.node_counts[2u * cid + 1u], 1ull); }
                                                tv_node = pass ? nd.info : nd.skip; };
#else
                        auto box_step = [&]() { const DFNode nd = fetch_fnode<ALL>(P, tv_node); tv_node = filter_pass(nd, F) ? nd.info : nd.skip; };
#endif'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)
# host: buffer + debug entry
p=root+'rt_host.cpp'; s=open(p).read()
old="    HIP_OK(hipMemsetAsync(slot->d_queue, 0, 64, stream));"
new='''#ifdef RT_COUNT_NODES
    {   // experiment build only (tools/passrate_probe.py): RT_NODE_COUNTS switches the per-node counters on for this launch
        static void* d_counts = nullptr; static size_t n_counts = 0;
        P.node_counts = nullptr;
        if (std::getenv("RT_NODE_COUNTS") && !f.bvh.empty()) {
            if (n_counts < f.bvh.size()) { if (d_counts) (void)hipFree(d_counts); HIP_OK(hipMalloc(&d_counts, f.bvh.size() * 16)); n_counts = f.bvh.size(); }
            HIP_OK(hipMemsetAsync(d_counts, 0, f.bvh.size() * 16, stream));
            P.node_counts = (unsigned long long*)d_counts;
            s.d_node_counts = d_counts;
        }
    }
#endif
    HIP_OK(hipMemsetAsync(slot->d_queue, 0, 64, stream));'''
assert old in s; s=s.replace(old,new,1)
old="// Test aid (host only): the flattened object table,"
new='''#ifdef RT_COUNT_NODES
// experiment build only: {box tests, passes} per BVH node of the last launch that ran with RT_NODE_COUNTS set
int rt_debug_node_counts(rt_scene* sc, unsigned long long* out, uint32_t max_nodes) {
    if (!sc || !out || !sc->s.d_node_counts) return set_err("no counted launch");
    HIP_OK(hipDeviceSynchronize());
    const size_t n = std::min<size_t>(max_nodes, sc->s.flat.bvh.size());
    HIP_OK(hipMemcpy(out, sc->s.d_node_counts, n * 16, hipMemcpyDeviceToHost));
    return (int)n;
}
#endif
// Test aid (host only): the flattened object table,'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)
p=root+'rt_scene.h'; s=open(p).read()
old="    void invalidate() {"
new="#ifdef RT_COUNT_NODES\n    void* d_node_counts = nullptr;      // experiment build only (tools/passrate_probe.py)\n#endif\n    void invalidate() {"
assert old in s; s=s.replace(old,new,1); open(p,'w').write(s)
# flattener: contraction set from a file
p=root+'rt_flatten.cpp'; s=open(p).read()
old="    auto resolve = [&](uint32_t x) { while (x != 0xFFFFFFFFu && redirect[x] != x) x = redirect[x]; return x; };"
new='''#ifdef RT_COUNT_NODES
    if (const char* path = std::getenv("RT_COLLAPSE_SET")) {      // experiment build only (tools/passrate_probe.py): the nodes to take out, one id per line
        for (size_t i = 0; i < n; i++) redirect[i] = (uint32_t)i;
        if (FILE* fp = std::fopen(path, "r")) {
            unsigned long id;
            while (std::fscanf(fp, "%lu", &id) == 1)
                if (id < n && !(f.bvh[id].a & BVH_LEAF) && parent[id] != 0xFFFFFFFFu) redirect[id] = f.bvh[id].c;
            std::fclose(fp);
        }
    }
#endif
    auto resolve = [&](uint32_t x) { while (x != 0xFFFFFFFFu && redirect[x] != x) x = redirect[x]; return x; };'''
assert old in s; s=s.replace(old,new,1); open(p,'w').write(s)
print("patched")
