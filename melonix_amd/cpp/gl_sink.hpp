// gl_sink.hpp — where SpecCache gets its OpenGL entry points from.
//   default                : the real headers, exactly what the reference includes
//                            (spec-cache.hpp:7-11, texture.hpp:2-8).
//   MELONIX_AMD_NO_GL      : headless builds (tests, servers): the five GL calls SpecCache and
//                            Texture make are routed to functions the embedding program defines.
#pragma once
#ifndef MELONIX_AMD_NO_GL
#if defined(IMGUI_IMPL_OPENGL_ES2)
#include <SDL_opengles2.h>
#else
#include <SDL_opengl.h>
#endif
#else
typedef unsigned int GLuint;
typedef unsigned int GLenum;
typedef int GLint;
typedef int GLsizei;
#define GL_TEXTURE_1D 0x0DE0
#define GL_TEXTURE_MAG_FILTER 0x2800
#define GL_TEXTURE_MIN_FILTER 0x2801
#define GL_NEAREST 0x2600
#define GL_RGB 0x1907
#define GL_UNSIGNED_BYTE 0x1401
extern "C" {
void glGenTextures(GLsizei n, GLuint *textures);
void glDeleteTextures(GLsizei n, const GLuint *textures);
void glBindTexture(GLenum target, GLuint texture);
void glTexParameteri(GLenum target, GLenum pname, GLint param);
void glTexImage1D(GLenum target, GLint level, GLint internalFormat, GLsizei width, GLint border, GLenum format,
                  GLenum type, const void *pixels);
}
#endif
