// pybind_shim.cpp -- the drop-in pybind11 surface: module `propagation`, class `Graph`.
//
// Built to precompute/propagation<ext-suffix>.so so that the reference's
// `from precompute import propagation` (model.py:9, model_mag.py:8) resolves to this
// module unchanged.  Same exported names and positional signatures as the reference's
// binding (precompute/propagation.cpp:9-11):
//     Graph(indptr: int32[], indices: int32[], seed: int)
//     Graph.gfpush_omp(node_idx, row_idx, col_idx, value, coef, rmax, K) -> None
// All computation goes through the C ABI of include/grandplus.h (HIP on gfx950); this
// file only converts arguments and maps status codes to Python exceptions.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "grandplus.h"

namespace py = pybind11;

namespace {

[[noreturn]] void raise_status(int status) {
    std::string msg = gp_last_error();
    if (msg.empty()) msg = gp_strerror(status);
    switch (status) {
        case GP_ERR_NULL: case GP_ERR_INVALID_CSR: case GP_ERR_INVALID_SEED: case GP_ERR_INVALID_ARG:
            throw py::value_error(msg);
        case GP_ERR_NOMEM:
            PyErr_SetString(PyExc_MemoryError, msg.c_str());
            throw py::error_already_set();
        default:
            throw std::runtime_error(msg);
    }
}

// Output arrays must alias the caller's memory.  The reference takes array_t<T>& and
// silently converts a mismatching array into a temporary whose contents are discarded
// (SURVEY.md A.2 Q7); here that is a TypeError.
template <typename T>
T* checked_output(py::array& a, const char* name, py::ssize_t need) {
    if (!py::dtype::of<T>().is(a.dtype()) && !a.dtype().equal(py::dtype::of<T>()))
        throw py::type_error(std::string(name) + " has the wrong dtype (results would be lost)");
    if (!(a.flags() & py::array::c_style) || !a.writeable())
        throw py::type_error(std::string(name) + " must be C-contiguous and writeable");
    if (a.size() < need)
        throw py::value_error(std::string(name) + " is shorter than len(node_idx) * K");
    return static_cast<T*>(a.mutable_data());
}

class Graph {
public:
    Graph(py::array_t<int, py::array::c_style | py::array::forcecast> indptr,
          py::array_t<int, py::array::c_style | py::array::forcecast> indices, int /*seed: unused, graph.h:40*/)
    {
        if (indptr.size() < 1) throw py::value_error("indptr must have at least one element");
        // The reference's caller is one process making one call (model.py:251, :268).  By default that call runs on ONE GPU
        // (GRANDPLUS_DEVICE / LOCAL_RANK, else device 0).  GRANDPLUS_GPUS=N (N > 1, or "all") makes it a multi-GPU handle over the
        // first N visible GPUs: gfpush_omp then shards its seeds over them inside the one call (small calls stay on the first GPU).
        // (Until round 5 the multi-GPU handle was the default on a multi-GPU node; its >= 2-part branch had then never run on
        // hardware -- VERDICT r4 -- so it is opt-in now.  GRANDPLUS_MULTI_DEVICES="0,0" builds the handle over an explicit device
        // list, repeats allowed: the sharded path on a one-GPU box.)
        const char* pin = std::getenv("GRANDPLUS_DEVICE");
        if (!pin) pin = std::getenv("LOCAL_RANK");
        const int ndev = gp_device_count();
        int n_gpus = 1;
        if (const char* e = std::getenv("GRANDPLUS_GPUS")) n_gpus = std::string(e) == "all" ? ndev : std::atoi(e);
        if (n_gpus < 1 || n_gpus > ndev) n_gpus = ndev > 0 ? ndev : 1;
        std::vector<int> devs;
        if (const char* e = std::getenv("GRANDPLUS_MULTI_DEVICES")) {
            std::string s(e);
            size_t pos = 0;
            while (pos < s.size()) {
                size_t next = s.find(',', pos);
                if (next == std::string::npos) next = s.size();
                if (next > pos) devs.push_back(std::atoi(s.substr(pos, next - pos).c_str()));
                pos = next + 1;
            }
        }
        int rc;
        {
            py::gil_scoped_release nogil;
            if (!devs.empty()) {
                rc = gp_graph_create_multi_on(indptr.data(), indptr.size() - 1, indices.data(), indices.size(), devs.data(), (int)devs.size(), &g_);
            } else if (pin || n_gpus <= 1) {
                int device = pin ? std::atoi(pin) : 0;
                if (device < 0 || device >= ndev) device = 0;
                rc = gp_graph_create(indptr.data(), indptr.size() - 1, indices.data(), indices.size(), device, &g_);
            } else {
                rc = gp_graph_create_multi(indptr.data(), indptr.size() - 1, indices.data(), indices.size(), n_gpus, &g_);
            }
        }
        if (rc != GP_OK) raise_status(rc);
    }
    ~Graph() { gp_graph_destroy(g_); }
    Graph(const Graph&) = delete;
    Graph& operator=(const Graph&) = delete;

    void gfpush_omp(py::array_t<int, py::array::c_style | py::array::forcecast> node_idx,
                    py::array row_idx, py::array col_idx, py::array value,
                    py::array_t<double, py::array::c_style | py::array::forcecast> coef,
                    double rmax, int K)
    {
        if (K < 1) throw py::value_error("K must be >= 1");
        const py::ssize_t S = node_idx.size();
        const py::ssize_t need = S * (py::ssize_t)K;
        int* row = checked_output<int>(row_idx, "row_idx", need);
        int* col = checked_output<int>(col_idx, "col_idx", need);
        double* val = checked_output<double>(value, "value", need);
        int rc;
        {
            py::gil_scoped_release nogil;        // the reference holds the GIL for the whole call
            rc = gp_gfpush(g_, node_idx.data(), S, coef.data(), (int)coef.size(), rmax, K, row, col, val);
        }
        if (rc != GP_OK) raise_status(rc);
    }

private:
    gp_graph* g_ = nullptr;
};

}  // namespace

PYBIND11_MODULE(propagation, m) {
    m.doc() = "MI355X-native GFPush (drop-in for GRAND+'s precompute.propagation)";
    py::class_<Graph>(m, "Graph")
        .def(py::init<py::array_t<int, py::array::c_style | py::array::forcecast>,
                      py::array_t<int, py::array::c_style | py::array::forcecast>, int>())
        .def("gfpush_omp", &Graph::gfpush_omp);
}
