// oracle/ref_kdtree_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// The reference's own kd-tree over the DiSCO signatures (Mapping/src/global_manager/src/kdtree.cpp, compiled in place by
// oracle/Makefile into oracle/_ref/libref_kdtree.so), driven the way GlobalManager does it: kdtree_init(dim),
// kdtree_insert per descriptor (global_manager.cpp:1880-1884), kdtree_build + kdtree_knn_search + kdtree_knn_result per query
// (global_manager.cpp:1002-1007), knn_list_reset afterwards (:1185).
#include <cstring>
#include <vector>

#include "global_manager/kdtree.h"

extern "C" {

// db [n][dim], query [dim]; writes at most k (index, distance) pairs, nearest first; distance = sqrt(sum of squares) as the
// reference reports it.  Returns how many the tree returned.
int ref_kdtree_knn(const float* db, int n, int dim, const float* query, int k, long* index_out, float* distance_out)
{
    struct kdtree* tree = kdtree_init(dim);
    std::vector<float> row(dim);
    for (int i = 0; i < n; ++i) {
        std::memcpy(row.data(), db + (size_t)i * dim, sizeof(float) * dim);
        kdtree_insert(tree, row.data());
    }
    kdtree_build(tree);
    std::vector<float> q(query, query + dim);
    kdtree_knn_search(tree, q.data(), k);
    const std::vector<kdresult_t> res = kdtree_knn_result(tree);
    int m = 0;
    for (const kdresult_t& r : res) {
        if (m >= k) break;
        index_out[m] = r.coord_index;
        distance_out[m] = r.distance;
        ++m;
    }
    knn_list_reset(tree);
    kdtree_destroy(tree);
    return (int)res.size();
}

}  // extern "C"
