/*
 * lp_harness -- runs the reference's OWN fragment shader, read at run time
 * from $PT_REFERENCE_DIR/src/shaders/pathtracer/, on Mesa llvmpipe through the
 * DRI software-rasteriser interface (no X server, no window system), and
 * dumps what it rendered.  This is how the golden fixtures under
 * tests/golden/ were produced and how the oracle in ../pt_oracle.c is pinned.
 *
 * TEST INFRASTRUCTURE ONLY, and container-only: it needs the reference tree
 * and Mesa's swrast_dri.so.  Nothing of the reference is copied: the GLSL text
 * is loaded from where it lies and patched in memory (loop bound, NaN guard,
 * or a debug main() appended in place of the shader's own).
 *
 * The GL host side here is NOT the reference's renderer.cpp (that file needs
 * GLEW/GLFW, absent from the image); it re-creates the buffer-texture and
 * uniform contract renderer.cpp:11-112 sets up and the draw of
 * renderer.cpp:124-137, from arrays handed over in a flat binary file.
 *
 *   lp_harness SCENE.bin OUT.bin WIDTH HEIGHT MODE ITERS DEPTH
 *     MODE: ids      one pass, RGBA32F target, main() replaced: (dir.xyz, hit code)
 *           tn       one pass, main() replaced: (t, normal.xyz) of the primary hit
 *           rgba8    as shipped: RGBA8 feedback accumulation, ITERS iterations
 *           float    RGBA32F accumulation target, shader untouched
 *           float0   RGBA32F target + non-finite pass colour replaced by 0
 *           post     float0, then the reference's post pass (post.frag.glsl) drawn once more
 *                    into an RGBA8 texture; OUT.bin = the float image followed by
 *                    WIDTH*HEIGHT*4 bytes of what the post pass drew
 *     env: PT_REFERENCE_DIR (default /root/reference)
 *          MESA_GL_VERSION_OVERRIDE=4.6 MESA_GLSL_VERSION_OVERRIDE=460 are set
 *          here if absent (llvmpipe stops at 4.5 and rejects "#version 460").
 *   Output: WIDTH*HEIGHT*4 floats (or bytes for rgba8), row 0 = bottom row;
 *   stdout: one line "seconds=<render wall time>".
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

/* ---- GL entry points ---- */
#define GL_FUNCS(X)                                        \
    X(PFNGLGETERRORPROC, glGetError)                       \
    X(PFNGLGETSTRINGPROC, glGetString)                     \
    X(PFNGLGETINTEGERVPROC, glGetIntegerv)                 \
    X(PFNGLVIEWPORTPROC, glViewport)                       \
    X(PFNGLCREATESHADERPROC, glCreateShader)               \
    X(PFNGLSHADERSOURCEPROC, glShaderSource)               \
    X(PFNGLCOMPILESHADERPROC, glCompileShader)             \
    X(PFNGLGETSHADERIVPROC, glGetShaderiv)                 \
    X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog)       \
    X(PFNGLCREATEPROGRAMPROC, glCreateProgram)             \
    X(PFNGLATTACHSHADERPROC, glAttachShader)               \
    X(PFNGLLINKPROGRAMPROC, glLinkProgram)                 \
    X(PFNGLGETPROGRAMIVPROC, glGetProgramiv)               \
    X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog)     \
    X(PFNGLUSEPROGRAMPROC, glUseProgram)                   \
    X(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation)   \
    X(PFNGLUNIFORM1IPROC, glUniform1i)                     \
    X(PFNGLUNIFORM1UIPROC, glUniform1ui)                   \
    X(PFNGLUNIFORM2UIVPROC, glUniform2uiv)                 \
    X(PFNGLUNIFORM4UIVPROC, glUniform4uiv)                 \
    X(PFNGLUNIFORM3FVPROC, glUniform3fv)                   \
    X(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays)         \
    X(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray)         \
    X(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers)         \
    X(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer)         \
    X(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D) \
    X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus) \
    X(PFNGLDRAWBUFFERSPROC, glDrawBuffers)                 \
    X(PFNGLACTIVETEXTUREPROC, glActiveTexture)             \
    X(PFNGLGENTEXTURESPROC, glGenTextures)                 \
    X(PFNGLBINDTEXTUREPROC, glBindTexture)                 \
    X(PFNGLTEXIMAGE2DPROC, glTexImage2D)                   \
    X(PFNGLTEXPARAMETERIPROC, glTexParameteri)             \
    X(PFNGLGETTEXIMAGEPROC, glGetTexImage)                 \
    X(PFNGLGENBUFFERSPROC, glGenBuffers)                   \
    X(PFNGLBINDBUFFERPROC, glBindBuffer)                   \
    X(PFNGLBUFFERDATAPROC, glBufferData)                   \
    X(PFNGLTEXBUFFERPROC, glTexBuffer)                     \
    X(PFNGLDRAWARRAYSPROC, glDrawArrays)                   \
    X(PFNGLCLEARPROC, glClear)                             \
    X(PFNGLFINISHPROC, glFinish)

#define DECL(type, name) static type name;
GL_FUNCS(DECL)

static int g_width, g_height;

/* ---- DRI swrast loader callbacks: an off-screen drawable of fixed size ---- */
static void ld_get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *priv)
{
    (void)d; (void)priv;
    *x = 0; *y = 0; *w = g_width; *h = g_height;
}
static void ld_put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *priv)
{
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)priv;
}
static void ld_get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *priv)
{
    (void)d; (void)x; (void)y; (void)priv;
    memset(data, 0, (size_t)w * h * 4);
}
static void ld_put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *priv)
{
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)priv;
}
static void ld_get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *priv)
{
    (void)d; (void)x; (void)y; (void)w; (void)priv;
    memset(data, 0, (size_t)stride * h);
}

static __DRIswrastLoaderExtension g_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = ld_get_drawable_info,
    .putImage = ld_put_image,
    .getImage = ld_get_image,
    .putImage2 = ld_put_image2,
    .getImage2 = ld_get_image2,
};
static const __DRIextension *g_loader_exts[] = {&g_loader.base, NULL};

static void die(const char *msg)
{
    fprintf(stderr, "lp_harness: %s\n", msg);
    exit(1);
}

static char *read_text(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *s = malloc((size_t)n + 1);
    if (fread(s, 1, (size_t)n, f) != (size_t)n) die("short read");
    s[n] = 0;
    fclose(f);
    return s;
}

/* Returns a new string with the first occurrence of `from` replaced by `to`. */
static char *replace_once(const char *s, const char *from, const char *to)
{
    const char *p = strstr(s, from);
    if (!p) { fprintf(stderr, "pattern not found: %s\n", from); exit(1); }
    size_t n = strlen(s) + strlen(to) + 1;
    char *out = malloc(n);
    size_t head = (size_t)(p - s);
    memcpy(out, s, head);
    strcpy(out + head, to);
    strcat(out, p + strlen(from));
    return out;
}

static GLuint compile(GLenum type, const char *src)
{
    GLuint sh = glCreateShader(type);
    glShaderSource(sh, 1, &src, NULL);
    glCompileShader(sh);
    GLint ok = 0;
    glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[8192];
        glGetShaderInfoLog(sh, sizeof log, NULL, log);
        fprintf(stderr, "shader compile failed:\n%s\n", log);
        exit(1);
    }
    return sh;
}

/* Debug replacements for the shader's main(): everything above main() stays
 * the reference's text. */
static const char *MAIN_IDS =
    "void main()\n{\n"
    "    seed = uvec2(iterationCount + 1, iterationCount + 2) * uvec2(gl_FragCoord.xy);\n"
    "    Ray ray = raycast();\n"
    "    Intersection isect;\n"
    "    float code = -1.f;\n"
    "    if (intersect(ray, isect)) code = (isect.type == LIGHT) ? -(2.f + float(isect.index)) : float(isect.index);\n"
    "    out_color = vec4(ray.direction, code);\n"
    "}\n";
static const char *MAIN_TN =
    "void main()\n{\n"
    "    seed = uvec2(iterationCount + 1, iterationCount + 2) * uvec2(gl_FragCoord.xy);\n"
    "    Ray ray = raycast();\n"
    "    Intersection isect;\n"
    "    isect.normal = vec3(0.f);\n"
    "    bool hit = intersect(ray, isect);\n"
    "    out_color = vec4(hit ? isect.t : -1.f, isect.normal);\n"
    "}\n";

typedef struct {
    uint32_t n_vertices, n_triangles, n_lights, n_materials;
} scene_header;

static GLuint buffer_texture(GLenum unit, GLenum format, const void *data, size_t bytes)
{
    GLuint buf, tex;
    glActiveTexture(unit);
    glGenBuffers(1, &buf);
    glBindBuffer(GL_TEXTURE_BUFFER, buf);
    glBufferData(GL_TEXTURE_BUFFER, (GLsizeiptr)bytes, data, GL_STATIC_DRAW);
    glGenTextures(1, &tex);
    glBindTexture(GL_TEXTURE_BUFFER, tex);
    glTexBuffer(GL_TEXTURE_BUFFER, format, buf);
    return tex;
}

int main(int argc, char **argv)
{
    if (argc != 8) {
        fprintf(stderr, "usage: %s SCENE.bin OUT.bin WIDTH HEIGHT ids|tn|rgba8|float|float0|post ITERS DEPTH\n", argv[0]);
        return 2;
    }
    const char *scene_path = argv[1], *out_path = argv[2], *mode = argv[5];
    g_width = atoi(argv[3]);
    g_height = atoi(argv[4]);
    const int iters = atoi(argv[6]), depth = atoi(argv[7]);
    const int debug_pass = !strcmp(mode, "ids") || !strcmp(mode, "tn");
    const int post_dump = !strcmp(mode, "post");
    const int float_target = debug_pass || !strcmp(mode, "float") || !strcmp(mode, "float0") || post_dump;
    if (!debug_pass && !post_dump && strcmp(mode, "rgba8") && strcmp(mode, "float") && strcmp(mode, "float0")) die("bad mode");
    setenv("MESA_GL_VERSION_OVERRIDE", "4.6", 0);
    setenv("MESA_GLSL_VERSION_OVERRIDE", "460", 0);
    const char *ref = getenv("PT_REFERENCE_DIR");
    if (!ref) ref = "/root/reference";

    /* ---- scene arrays ---- */
    FILE *sf = fopen(scene_path, "rb");
    if (!sf) die("cannot open scene file");
    scene_header hd;
    if (fread(&hd, sizeof hd, 1, sf) != 1) die("bad scene file");
    size_t nv = hd.n_vertices, nt = hd.n_triangles, nl = hd.n_lights, nm = hd.n_materials;
    float *vertices = malloc(nv * 12 + 16), *indices = malloc(nt * 16 + 16), *vdata = malloc(nt * 72 + 16);
    float *lights = malloc(nl * 80 + 16), *materials = malloc(nm * 32 + 16);
    uint32_t *matmap = malloc(nt * 4 + 16);
    float cam[12];
    if (fread(vertices, 12, nv, sf) != nv || fread(indices, 16, nt, sf) != nt || fread(vdata, 72, nt, sf) != nt ||
        fread(lights, 80, nl, sf) != nl || fread(materials, 32, nm, sf) != nm || fread(matmap, 4, nt, sf) != nt ||
        fread(cam, 4, 12, sf) != 12)
        die("truncated scene file");
    fclose(sf);

    /* ---- llvmpipe context through DRI swrast ---- */
    if (!dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL)) die(dlerror());
    void *drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) die(dlerror());
    const __DRIextension **(*get_exts)(void) =
        (const __DRIextension **(*)(void))dlsym(drv, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
    if (!get_exts) die("no __driDriverGetExtensions_swrast");
    const __DRIextension **drv_exts = get_exts();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *swrast = NULL;
    for (int i = 0; drv_exts[i]; ++i) {
        if (!strcmp(drv_exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)drv_exts[i];
        if (!strcmp(drv_exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)drv_exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) die("DRI_Core / DRI_SWRast v4 not found");
    const __DRIconfig **configs = NULL;
    __DRIscreen *screen = swrast->createNewScreen2(0, g_loader_exts, drv_exts, &configs, NULL);
    if (!screen || !configs) die("createNewScreen2 failed");
    const __DRIconfig *cfg = NULL;
    for (int i = 0; configs[i]; ++i) {
        unsigned r = 0, a = 0, db = 1;
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_ALPHA_SIZE, &a);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        if (r == 8 && a == 8 && !db) { cfg = configs[i]; break; }
    }
    if (!cfg) cfg = configs[0];
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 6};
    unsigned err = 0;
    __DRIcontext *ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, cfg, NULL, 2, attribs, &err, NULL);
    if (!ctx) { fprintf(stderr, "createContextAttribs failed: %u\n", err); return 1; }
    __DRIdrawable *draw = swrast->createNewDrawable(screen, cfg, NULL);
    if (!draw) die("createNewDrawable failed");
    if (!core->bindContext(ctx, draw, draw)) die("bindContext failed");

    void *(*get_proc)(const char *) = (void *(*)(const char *))dlsym(RTLD_DEFAULT, "_glapi_get_proc_address");
    if (!get_proc) die("no _glapi_get_proc_address");
#define LOAD(type, name)                                    \
    name = (type)get_proc(#name);                           \
    if (!name) die("missing GL entry point " #name);
    GL_FUNCS(LOAD)
    fprintf(stderr, "GL_RENDERER: %s\nGL_VERSION: %s\n", glGetString(GL_RENDERER), glGetString(GL_VERSION));
    {
        /* the scene arrays are buffer textures: a scene with more texels than this in one of them (6 per triangle in
           the vertex-data texture) is not rendered as the shader means it -- fetches past the limit read zeros */
        GLint max_texels = 0;
        glGetIntegerv(GL_MAX_TEXTURE_BUFFER_SIZE, &max_texels);
        fprintf(stderr, "GL_MAX_TEXTURE_BUFFER_SIZE: %d texels\n", max_texels);
    }

    /* ---- the reference's shaders, patched in memory ---- */
    char path[4096];
    snprintf(path, sizeof path, "%s/src/shaders/pathtracer/pathtracer.vert.glsl", ref);
    char *vert = read_text(path);
    snprintf(path, sizeof path, "%s/src/shaders/pathtracer/pathtracer.frag.glsl", ref);
    char *frag = read_text(path);
    if (debug_pass) {
        char *m = strstr(frag, "void main()");
        if (!m) die("main() not found in the fragment shader");
        *m = 0;
        const char *tail = !strcmp(mode, "ids") ? MAIN_IDS : MAIN_TN;
        char *joined = malloc(strlen(frag) + strlen(tail) + 1);
        strcpy(joined, frag);
        strcat(joined, tail);
        frag = joined;
    } else {
        if (depth != 10) {
            char bound[64];
            snprintf(bound, sizeof bound, "i < %d;", depth);
            frag = replace_once(frag, "i < 10;", bound);
        }
        if (!strcmp(mode, "float0") || post_dump)
            frag = replace_once(frag, "vec3 passCol = attenuation * intensity;",
                                "vec3 passCol = attenuation * intensity;\n"
                                "    if (any(isnan(passCol)) || any(isinf(passCol))) passCol = vec3(0.f);");
    }
    GLuint prog = glCreateProgram();
    glAttachShader(prog, compile(GL_VERTEX_SHADER, vert));
    glAttachShader(prog, compile(GL_FRAGMENT_SHADER, frag));
    glLinkProgram(prog);
    GLint ok = 0;
    glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[8192];
        glGetProgramInfoLog(prog, sizeof log, NULL, log);
        fprintf(stderr, "link failed:\n%s\n", log);
        return 1;
    }

    /* The reference's post pass (renderer.cpp:137-147) runs after every
     * accumulation pass and samples the accumulation texture; it is kept here
     * because it is part of one Renderer::draw() and because that read is what
     * orders consecutive feedback passes on llvmpipe. */
    snprintf(path, sizeof path, "%s/src/shaders/pathtracer/post.frag.glsl", ref);
    char *post = read_text(path);
    GLuint post_prog = glCreateProgram();
    glAttachShader(post_prog, compile(GL_VERTEX_SHADER, vert));
    glAttachShader(post_prog, compile(GL_FRAGMENT_SHADER, post));
    glLinkProgram(post_prog);
    glGetProgramiv(post_prog, GL_LINK_STATUS, &ok);
    if (!ok) die("post program link failed");

    GLuint vao;
    glGenVertexArrays(1, &vao);
    glBindVertexArray(vao);

    /* ---- accumulation target (renderer.cpp:15-30) ---- */
    GLuint fbo, accum;
    glGenFramebuffers(1, &fbo);
    glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    glActiveTexture(GL_TEXTURE3);
    glGenTextures(1, &accum);
    glBindTexture(GL_TEXTURE_2D, accum);
    if (float_target)
        glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, g_width, g_height, 0, GL_RGBA, GL_FLOAT, NULL);
    else
        glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, g_width, g_height, 0, GL_RGB, GL_UNSIGNED_BYTE, NULL);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, accum, 0);
    GLenum bufs[1] = {GL_COLOR_ATTACHMENT0};
    glDrawBuffers(1, bufs);
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("framebuffer incomplete");
    glBindFramebuffer(GL_FRAMEBUFFER, 0);

    /* ---- scene buffer textures, units and formats of renderer.cpp:32-83 ---- */
    buffer_texture(GL_TEXTURE0, GL_RGB32F, vertices, nv * 12);
    buffer_texture(GL_TEXTURE1, GL_RGBA32UI, indices, nt * 16); /* floats under an integer format, as the reference does */
    buffer_texture(GL_TEXTURE2, GL_RGB32F, vdata, nt * 72);
    buffer_texture(GL_TEXTURE4, GL_RGBA32F, lights, nl * 80);
    buffer_texture(GL_TEXTURE5, GL_R32F, materials, nm * 32);
    buffer_texture(GL_TEXTURE6, GL_R32UI, matmap, nt * 4);

    /* ---- uniforms (renderer.cpp:86-107, 189-199) ---- */
    glUseProgram(prog);
    GLuint light_count[4] = {(GLuint)nl, 0, 0, 0};
    GLuint res[2] = {(GLuint)g_width, (GLuint)g_height};
    glUniform1ui(glGetUniformLocation(prog, "indexCount"), (GLuint)nt);
    glUniform1i(glGetUniformLocation(prog, "verticesTex"), 0);
    glUniform1i(glGetUniformLocation(prog, "indicesTex"), 1);
    glUniform1i(glGetUniformLocation(prog, "vertexDataTex"), 2);
    glUniform1i(glGetUniformLocation(prog, "accumTexture"), 3);
    glUniform1i(glGetUniformLocation(prog, "lightTex"), 4);
    glUniform4uiv(glGetUniformLocation(prog, "lightCount"), 1, light_count);
    glUniform1i(glGetUniformLocation(prog, "materialTex"), 5);
    glUniform1i(glGetUniformLocation(prog, "materialMapTex"), 6);
    glUniform3fv(glGetUniformLocation(prog, "eye"), 1, cam + 0);
    glUniform3fv(glGetUniformLocation(prog, "forward"), 1, cam + 3);
    glUniform3fv(glGetUniformLocation(prog, "up"), 1, cam + 6);
    glUniform3fv(glGetUniformLocation(prog, "right"), 1, cam + 9);
    glUniform2uiv(glGetUniformLocation(prog, "resolution"), 1, res);
    glUseProgram(post_prog);
    glUniform1i(glGetUniformLocation(post_prog, "inTexture"), 3);
    glUseProgram(0);

    /* ---- draws (renderer.cpp:128-137) ---- */
    glViewport(0, 0, g_width, g_height);
    glFinish();
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int first = debug_pass ? depth : 0; /* debug passes: DEPTH argument carries the iteration index */
    const int count = debug_pass ? 1 : iters;
    for (int it = first; it < first + count; ++it) {
        glBindFramebuffer(GL_FRAMEBUFFER, fbo);
        glUseProgram(prog);
        glUniform1ui(glGetUniformLocation(prog, "iterationCount"), (GLuint)it);
        glDrawArrays(GL_TRIANGLES, 0, 3);
        glBindFramebuffer(GL_FRAMEBUFFER, 0);
        if (!debug_pass) {
            glViewport(0, 0, g_width, g_height);
            glClear(GL_COLOR_BUFFER_BIT);
            glUseProgram(post_prog);
            glActiveTexture(GL_TEXTURE3);
            glBindTexture(GL_TEXTURE_2D, accum);
            glDrawArrays(GL_TRIANGLES, 0, 3);
            glUseProgram(0);
        }
    }
    glFinish();
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (glGetError() != GL_NO_ERROR) die("GL error during rendering");

    /* ---- read back the accumulation texture (still bound to unit 3) ---- */
    glActiveTexture(GL_TEXTURE3);
    glBindTexture(GL_TEXTURE_2D, accum);
    size_t bytes = (size_t)g_width * g_height * 4 * (float_target ? 4 : 1);
    void *pixels = malloc(bytes);
    glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, float_target ? GL_FLOAT : GL_UNSIGNED_BYTE, pixels);
    FILE *of = fopen(out_path, "wb");
    if (!of || fwrite(pixels, 1, bytes, of) != bytes) die("cannot write output");
    if (post_dump) {
        /* The post pass of renderer.cpp:141-147 once more, into a texture that can be read back (the
         * reference draws it to the window's RGBA8 back buffer). */
        GLuint pfbo, ptex;
        glGenFramebuffers(1, &pfbo);
        glBindFramebuffer(GL_FRAMEBUFFER, pfbo);
        glActiveTexture(GL_TEXTURE7);
        glGenTextures(1, &ptex);
        glBindTexture(GL_TEXTURE_2D, ptex);
        glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, g_width, g_height, 0, GL_RGBA, GL_UNSIGNED_BYTE, NULL);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
        glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, ptex, 0);
        glDrawBuffers(1, bufs);
        if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("post framebuffer incomplete");
        glViewport(0, 0, g_width, g_height);
        glClear(GL_COLOR_BUFFER_BIT);
        glUseProgram(post_prog);
        glActiveTexture(GL_TEXTURE3);
        glBindTexture(GL_TEXTURE_2D, accum);
        glDrawArrays(GL_TRIANGLES, 0, 3);
        glUseProgram(0);
        glFinish();
        glActiveTexture(GL_TEXTURE7);
        glBindTexture(GL_TEXTURE_2D, ptex);
        size_t pbytes = (size_t)g_width * g_height * 4;
        void *ppix = malloc(pbytes);
        glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_UNSIGNED_BYTE, ppix);
        if (glGetError() != GL_NO_ERROR) die("GL error in the post dump");
        if (fwrite(ppix, 1, pbytes, of) != pbytes) die("cannot write output");
    }
    fclose(of);
    printf("seconds=%.6f\n", (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
    return 0;
}
