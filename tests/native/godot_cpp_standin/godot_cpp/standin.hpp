// A STAND-IN for the handful of godot-cpp 4.3 declarations shim/gsplat_gdextension.cpp uses — NOT godot-cpp.  godot-cpp
// is not in the build image, so without this the GDExtension class would never meet a compiler; with it the translation
// unit is parsed, type-checked (-Wall -Wextra -Werror) and — the stand-ins being small working classes — RUN through a
// session (tests/native/gdext_driver.cpp).  Signatures are godot-cpp's as far as the shim touches them (names, constness,
// argument and return types); everything else about the real classes is absent.  A build in a real godot-cpp tree remains
// the maintainer's step.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace godot {

typedef float real_t;

struct Vector2 {
    real_t x = 0, y = 0;
    Vector2() {}
    Vector2(real_t p_x, real_t p_y) : x(p_x), y(p_y) {}
};
struct Vector3 {
    real_t x = 0, y = 0, z = 0;
    Vector3() {}
    Vector3(real_t p_x, real_t p_y, real_t p_z) : x(p_x), y(p_y), z(p_z) {}
};
struct Basis {
    Vector3 rows[3] = {Vector3(1, 0, 0), Vector3(0, 1, 0), Vector3(0, 0, 1)};
    Vector3 get_column(int p_index) const {
        const real_t *r0 = &rows[0].x, *r1 = &rows[1].x, *r2 = &rows[2].x;
        return Vector3(r0[p_index], r1[p_index], r2[p_index]);
    }
};
struct Transform3D {
    Basis basis;
    Vector3 origin;
};

class String {
    std::string s;

public:
    String() {}
    String(const char *p) : s(p) {}
    const std::string &std_string() const { return s; }
    bool operator<(const String &o) const { return s < o.s; }
};
typedef String StringName;

template <class T>
class PackedArray {
    std::vector<T> v;

public:
    int64_t size() const { return (int64_t)v.size(); }
    int64_t resize(int64_t p_size) { v.resize((size_t)p_size); return 0; }
    const T *ptr() const { return v.data(); }
    T *ptrw() { return v.data(); }
};
typedef PackedArray<uint8_t> PackedByteArray;
typedef PackedArray<float> PackedFloat32Array;

class Variant {
public:
    enum Type { NIL, BOOL, INT, FLOAT, STRING };
    Type type = NIL;
    int64_t i = 0;
    double f = 0;
    std::string s;
    Variant() {}
    Variant(bool v) : type(BOOL), i(v) {}
    Variant(int32_t v) : type(INT), i(v) {}
    Variant(int64_t v) : type(INT), i(v) {}
    Variant(float v) : type(FLOAT), f(v) {}
    Variant(double v) : type(FLOAT), f(v) {}
    Variant(const char *v) : type(STRING), s(v) {}
    bool operator<(const Variant &o) const { return s < o.s; }
};
class Dictionary {
    std::map<Variant, Variant> m;

public:
    Variant &operator[](const Variant &p_key) { return m[p_key]; }
    int64_t size() const { return (int64_t)m.size(); }
    bool has(const Variant &p_key) const { return m.count(p_key) != 0; }
};

// what the stand-in ClassDB / Object record, for the driver to look at
struct StandinRegistry {
    std::vector<std::string> classes, methods, signals, emitted;
    static StandinRegistry &get() {
        static StandinRegistry r;
        return r;
    }
};

struct MethodDefinition {
    StringName name;
    std::vector<StringName> args;
};
template <class... Args>
MethodDefinition D_METHOD(const char *p_name, Args... p_args) {
    MethodDefinition d;
    d.name = p_name;
    d.args = {StringName(p_args)...};
    return d;
}
struct MethodInfo {
    StringName name;
    MethodInfo(const char *p_name) : name(p_name) {}
};
class MethodBind {};

class Object {
public:
    virtual ~Object() {}
    template <class... Args>
    int emit_signal(const StringName &p_signal, const Args &.../*p_args*/) {
        StandinRegistry::get().emitted.push_back(p_signal.std_string());
        return 0;
    }
};
class RefCounted : public Object {};

class ClassDB {
public:
    template <class N, class M, typename... VarArgs>
    static MethodBind *bind_method(N p_method_name, M /*p_method*/, VarArgs... /*p_args*/) {
        static MethodBind b;
        StandinRegistry::get().methods.push_back(MethodDefinition(p_method_name).name.std_string());
        return &b;
    }
    static void add_signal(const StringName & /*p_class*/, const MethodInfo &p_signal) {
        StandinRegistry::get().signals.push_back(p_signal.name.std_string());
    }
    template <class T>
    static void register_class() {
        StandinRegistry::get().classes.push_back(T::get_class_static().std_string());
        T::_bind_methods();
    }
};

#define GDCLASS(m_class, m_inherits)                                   \
private:                                                               \
    friend class ::godot::ClassDB;                                     \
                                                                       \
public:                                                                \
    typedef m_inherits parent_class;                                   \
    static ::godot::StringName get_class_static() { return #m_class; } \
                                                                       \
private:

#define ADD_SIGNAL(m_signal) ::godot::ClassDB::add_signal(get_class_static(), m_signal)

class Node3D : public Object {
public:
    Transform3D standin_transform;
    Transform3D get_global_transform() const { return standin_transform; }
};
class Camera3D : public Node3D {
public:
    double standin_fov = 75.0, standin_near = 0.05, standin_far = 4000.0;
    double get_fov() const { return standin_fov; }
    double get_near() const { return standin_near; }
    double get_far() const { return standin_far; }
};
class Time : public Object {
public:
    uint64_t standin_ticks_msec = 0;
    static Time *get_singleton() {
        static Time t;
        return &t;
    }
    uint64_t get_ticks_msec() const { return standin_ticks_msec; }
};

// gdextension_interface.h / godot.hpp
enum ModuleInitializationLevel {
    MODULE_INITIALIZATION_LEVEL_CORE,
    MODULE_INITIALIZATION_LEVEL_SERVERS,
    MODULE_INITIALIZATION_LEVEL_SCENE,
    MODULE_INITIALIZATION_LEVEL_EDITOR,
};

}  // namespace godot

typedef uint8_t GDExtensionBool;
typedef void *GDExtensionClassLibraryPtr;
typedef void (*GDExtensionInterfaceFunctionPtr)();
typedef GDExtensionInterfaceFunctionPtr (*GDExtensionInterfaceGetProcAddress)(const char *p_function_name);
struct GDExtensionInitialization {
    int minimum_initialization_level = 0;
    void *userdata = nullptr;
    void (*initialize)(void *userdata, int p_level) = nullptr;
    void (*deinitialize)(void *userdata, int p_level) = nullptr;
};
#define GDE_EXPORT __attribute__((visibility("default")))

namespace godot {
class GDExtensionBinding {
public:
    typedef void (*Callback)(ModuleInitializationLevel p_level);
    class InitObject {
        GDExtensionInitialization *initialization;
        mutable Callback init_callback = nullptr, terminate_callback = nullptr;
        mutable ModuleInitializationLevel minimum_level = MODULE_INITIALIZATION_LEVEL_CORE;

    public:
        InitObject(GDExtensionInterfaceGetProcAddress, GDExtensionClassLibraryPtr, GDExtensionInitialization *r_initialization)
            : initialization(r_initialization) {}
        void register_initializer(Callback p_init) const { init_callback = p_init; }
        void register_terminator(Callback p_terminate) const { terminate_callback = p_terminate; }
        void set_minimum_library_initialization_level(ModuleInitializationLevel p_level) const { minimum_level = p_level; }
        GDExtensionBool init() const {   // (the engine would call the initializer level by level; the stand-in does it here)
            if (initialization) initialization->minimum_initialization_level = (int)minimum_level;
            for (int l = 0; l <= (int)MODULE_INITIALIZATION_LEVEL_EDITOR; ++l)
                if (init_callback) init_callback((ModuleInitializationLevel)l);
            return 1;
        }
    };
};
}  // namespace godot
