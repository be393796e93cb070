/*
 * element.hpp -- mesh carriers.  The reference keeps one Element<DType> object
 * per element with a dozen std::vectors each (element.hpp:18-55).  Here the
 * mesh of a rank at one degree is a struct of flat arrays in exactly the
 * layout Domain::initialize reads from disk (domain.tpp:45-224; element-major,
 * (N+1)^dim values per element, x fastest), and Element is a light view into
 * it with the reference's field names.
 */
#ifndef FDD_ELEMENT_HPP
#define FDD_ELEMENT_HPP

#include <cmath>
#include <vector>

#include "config.hpp"

template <typename DType>
struct MeshData
{
    int dim = 3;
    int poly_degree = 1;
    int num_local_elements = 0;
    int num_total_elements = 0; // filled by Domain::initialize (all-reduce)

    std::vector<DType> x, y, z;             // GLL coordinates
    std::vector<long long> glo_num;         // 1-based global node id
    std::vector<int> node_degree;           // global multiplicity per local point
    std::vector<DType> p_mask;              // 0 on Dirichlet boundary else 1
    std::vector<DType> g[NUM_GEOM_FACTS];   // geometric factors incl. quadrature weights

    int num_elem_points() const { return (int)std::lround(std::pow(poly_degree + 1, dim)); }
    int num_local_points() const { return num_local_elements * num_elem_points(); }
};

template <typename DType>
class Element
{
  public:
    int id = 0;
    int dim = 3;
    int poly_degree = 1;
    int num_points = 0;
    int offset = 0;
    int n_x = 0, n_y = 0, n_z = 0;

    // views into the owning MeshData
    const DType *x = nullptr, *y = nullptr, *z = nullptr;
    const DType *dirichlet_mask = nullptr;
    const DType *geom_fact[NUM_GEOM_FACTS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const long long *glo_num = nullptr;

    Element() {}
    Element(int id_, int dim_, int poly_degree_) : id(id_), dim(dim_), poly_degree(poly_degree_)
    {
        n_x = poly_degree + 1;
        n_y = poly_degree + 1;
        n_z = (dim == 3) ? poly_degree + 1 : 1;
        num_points = n_x * n_y * n_z;
        offset = id * num_points;
    }

    void bind(const MeshData<DType> &m)
    {
        x = m.x.empty() ? nullptr : m.x.data() + offset;
        y = m.y.empty() ? nullptr : m.y.data() + offset;
        z = m.z.empty() ? nullptr : m.z.data() + offset;
        dirichlet_mask = m.p_mask.data() + offset;
        glo_num = m.glo_num.data() + offset;
        for (int g = 0; g < NUM_GEOM_FACTS; g++) geom_fact[g] = m.g[g].empty() ? nullptr : m.g[g].data() + offset;
    }
};

#endif
