// reference_shapes.h -- TEST DOUBLES with the interface shapes of the reference's headers, in the global namespace
// and under the reference's names, so that integration/hip_module.h can be type-checked the way it is compiled inside
// Tracy: TracyModule / TracyEvent (src/module.h:12-54), WindowHandle / handle_t (src/common.h:292-300), Scene
// (src/scene.h:36-60), Mesh (mesh.h:83-103), Material (material.h:17,54-101), Texture (texture.h:59-82), Camera
// (camera.h:57-67), BBox (aabb.h:58-59). They hold no logic and are not a build of the reference (whose headers need
// the un-vendored cclib).
#pragma once
#include <cstdint>
#include <vector>

using u32 = uint32_t;
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct mat4 { float m[16]; };
struct BBox { vec3 minbound, maxbound; };
struct Vertex { float pos[3], normal[3], uv0[2], tangent[3], bitangent[3]; };
using Index = uint32_t;

struct handle_t { u32 width; u32 height; int ds; void* dpy; unsigned long win; };
using WindowHandle = handle_t*;

class Mesh {
public:
    u32 GetVertexCount() const { return 0; }
    u32 GetIndexCount() const { return 0; }
    const Vertex* GetVertices() const { return nullptr; }
    const Index* GetIndices() const { return nullptr; }
    u32 GetMaterial() const { return 0; }
    const BBox& GetAABB() const { return aabb_; }
private:
    BBox aabb_{};
};
class Material {
public:
    enum TextureID { eBASECOLOR, eNORMAL, eROUGHNESS, eMETALNESS, eEMISSIVE, eCOUNT };
    u32 GetTexture(TextureID) const { return UINT32_MAX; }
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
    u32 GetWidth() const { return 0; }
    u32 GetHeight() const { return 0; }
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
    const Mesh& GetObject(u32 i) const { return objects_[i]; }
    u32 GetObjectCount() const { return (u32)objects_.size(); }
    u32 GetWidth() const { return 0; }
    u32 GetHeight() const { return 0; }
    const std::vector<Material>& GetMaterials() const { return materials_; }
    const std::vector<Texture>& GetTextures() const { return textures_; }
private:
    Camera camera_;
    std::vector<Mesh> objects_;
    std::vector<Material> materials_;
    std::vector<Texture> textures_;
};

enum class TracyEvent { eCreate, eResize, eCameraCut, eDestroy };
class TracyModule {
public:
    TracyModule() {}
    virtual ~TracyModule() {}
    TracyModule(const TracyModule&) = delete;
    TracyModule& operator=(const TracyModule&) = delete;
    virtual bool Startup(const WindowHandle in_Window, const Scene& in_Scene) = 0;
    virtual void Shutdown() = 0;
    virtual void OnUpdate(const Scene& in_Scene, float in_DeltaTime) = 0;
    virtual void OnRender(const WindowHandle in_Window) = 0;
    virtual void OnEvent(TracyEvent in_Event, const WindowHandle in_Window, const Scene& in_Scene) = 0;
    virtual const char* GetModuleName() const = 0;
    virtual u32 GetRayCount(bool in_ShouldReset = false) { (void)in_ShouldReset; return 0; }
};
