// Compile-only check that the generic tracy_hip::FlattenScene / HipTraceT (hip_trace.h) type-check
// against a scene with the ACCESSOR SIGNATURES of the reference (src/scene.h:36-60, mesh.h:83-103,
// material.h:17,54-101, texture.h:59-82, camera.h:57-67, aabb.h:58-59). The classes below are test
// doubles holding no logic; they are not a build of the reference.
#include <cstdint>
#include <vector>

#include "../../tracy_amd/csrc/host/hip_trace.h"

namespace mock {
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct mat4 { float m[16]; };
struct BBox { vec3 minbound, maxbound; };
struct Vertex { float pos[3], normal[3], uv0[2], tangent[3], bitangent[3]; };
using Index = uint32_t;
class Mesh {
public:
    uint32_t GetVertexCount() const { return 0; }
    uint32_t GetIndexCount() const { return 0; }
    const Vertex* GetVertices() const { return nullptr; }
    const Index* GetIndices() const { return nullptr; }
    uint32_t GetMaterial() const { return 0; }
    const BBox& GetAABB() const { return aabb_; }
private:
    BBox aabb_{};
};
class Material {
public:
    enum TextureID { eBASECOLOR, eNORMAL, eROUGHNESS, eMETALNESS, eEMISSIVE, eCOUNT };
    uint32_t GetTexture(TextureID) const { return UINT32_MAX; }
    const vec3& GetAlbedo() const { return v_; }
    const vec3& GetEmissive() const { return v_; }
    float GetRoughness() const { return 0; }
    float GetMetalness() const { return 0; }
    float GetRefractiveIndex() const { return 1; }
    float GetTranslucent() const { return 0; }
private:
    vec3 v_{};
};
class Texture {
public:
    uint32_t GetWidth() const { return 0; }
    uint32_t GetHeight() const { return 0; }
    vec4* GetPixels() const { return nullptr; }
};
class Camera {
public:
    const mat4& GetView() const { return m_; }
    const mat4& GetProjection() const { return m_; }
    const vec3& GetPosition() const { return p_; }
private:
    mat4 m_{};
    vec3 p_{};
};
class Scene {
public:
    const Camera& GetCamera() const { return camera_; }
    const Mesh& GetObject(uint32_t i) const { return objects_[i]; }
    uint32_t GetObjectCount() const { return (uint32_t)objects_.size(); }
    uint32_t GetWidth() const { return 0; }
    uint32_t GetHeight() const { return 0; }
    const std::vector<Material>& GetMaterials() const { return materials_; }
    const std::vector<Texture>& GetTextures() const { return textures_; }
private:
    Camera camera_;
    std::vector<Mesh> objects_;
    std::vector<Material> materials_;
    std::vector<Texture> textures_;
};
}  // namespace mock

template class tracy_hip::HipTraceT<mock::Scene>;

int main() {
    tracy_hip::FlatScene flat;
    mock::Scene s;
    return tracy_hip::FlattenScene(s, flat) ? 0 : 1;
}
