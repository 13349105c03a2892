#include "image_view.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace mif
{
    namespace
    {
        bool read_token(std::istream& in, std::string& tok)
        {
            tok.clear();
            int ch;
            while ((ch = in.get()) != EOF) {
                if (ch == '#') {
                    while ((ch = in.get()) != EOF && ch != '\n') {}
                    continue;
                }
                if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r') {
                    if (!tok.empty()) return true;
                    continue;
                }
                tok.push_back((char)ch);
            }
            return !tok.empty();
        }
    }

    Image imread(const std::string& path, bool grayscale)
    {
        Image im;
        std::ifstream in(path, std::ios::binary);
        if (!in.is_open()) return im;
        std::string magic, sw, sh, smax;
        if (!read_token(in, magic) || !read_token(in, sw) || !read_token(in, sh) || !read_token(in, smax)) return im;
        const int ch = magic == "P5" ? 1 : magic == "P6" ? 3 : 0;
        const int w = std::atoi(sw.c_str()), h = std::atoi(sh.c_str());
        if (ch == 0 || w <= 0 || h <= 0 || std::atoi(smax.c_str()) != 255) return im;
        std::vector<unsigned char> buf((size_t)w * h * ch);
        in.read((char*)buf.data(), (std::streamsize)buf.size());
        if ((size_t)in.gcount() != buf.size()) return im;
        im.rows = h; im.cols = w;
        if (ch == 1) {
            im.nchannels = 1;
            im.storage.swap(buf);
        } else if (grayscale) {
            im.nchannels = 1;
            im.storage.resize((size_t)w * h);
            for (size_t i = 0; i < (size_t)w * h; i++) {
                const int r = buf[3 * i], g = buf[3 * i + 1], b = buf[3 * i + 2];
                im.storage[i] = (unsigned char)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
            }
        } else {
            im.nchannels = 3;
            im.storage.resize(buf.size());
            for (size_t i = 0; i < (size_t)w * h; i++) {
                im.storage[3 * i] = buf[3 * i + 2];
                im.storage[3 * i + 1] = buf[3 * i + 1];
                im.storage[3 * i + 2] = buf[3 * i];
            }
        }
        im.data = im.storage.data();
        return im;
    }

    bool imwrite(const std::string& path, const Image& image)
    {
        if (image.empty() || (image.nchannels != 1 && image.nchannels != 3)) return false;
        std::ofstream out(path, std::ios::binary);
        if (!out.is_open()) return false;
        out << (image.nchannels == 1 ? "P5" : "P6") << "\n" << image.cols << " " << image.rows << "\n255\n";
        const size_t n = (size_t)image.rows * image.cols;
        if (image.nchannels == 1) {
            out.write((const char*)image.data, (std::streamsize)n);
        } else {
            std::vector<unsigned char> rgb(n * 3);
            for (size_t i = 0; i < n; i++) {
                rgb[3 * i] = image.data[3 * i + 2];
                rgb[3 * i + 1] = image.data[3 * i + 1];
                rgb[3 * i + 2] = image.data[3 * i];
            }
            out.write((const char*)rgb.data(), (std::streamsize)rgb.size());
        }
        return out.good();
    }

    FrameReader::FrameReader(const std::string& spec)
    {
        const size_t colon = spec.rfind(':');
        int w = 0, h = 0, c = 0;
        if (colon != std::string::npos && std::sscanf(spec.c_str() + colon + 1, "%dx%dx%d", &w, &h, &c) == 3 &&
            w > 0 && h > 0 && (c == 1 || c == 3)) {
            raw_ = true;
            path_ = spec.substr(0, colon);
            w_ = w; h_ = h; c_ = c;
            std::ifstream probe(path_, std::ios::binary);
            opened_ = probe.is_open();
        } else {
            path_ = spec;
            char name[4096];
            std::snprintf(name, sizeof(name), path_.c_str(), 0);
            std::ifstream probe(name, std::ios::binary);
            opened_ = probe.is_open();
        }
    }

    size_t FrameReader::count(size_t limit) const
    {
        if (!opened_) return 0;
        if (raw_) {
            std::ifstream in(path_, std::ios::binary | std::ios::ate);
            const size_t bytes = (size_t)w_ * h_ * c_;
            const size_t n = in.is_open() ? (size_t)in.tellg() / bytes : 0;
            return n < limit ? n : limit;
        }
        size_t n = 0;
        for (; n < limit; n++) {
            char name[4096];
            std::snprintf(name, sizeof(name), path_.c_str(), (int)n);
            if (n > 0 && std::string(name) == path_) break;                // a single file, not a pattern
            std::ifstream probe(name, std::ios::binary);
            if (!probe.is_open()) break;
        }
        return n;
    }

    bool FrameReader::read(Image& frame)
    {
        if (!opened_) return false;
        if (raw_) {
            const size_t bytes = (size_t)w_ * h_ * c_;
            std::ifstream in(path_, std::ios::binary);
            in.seekg((std::streamoff)(index_ * bytes));
            frame = Image();
            frame.storage.resize(bytes);
            in.read((char*)frame.storage.data(), (std::streamsize)bytes);
            if ((size_t)in.gcount() != bytes) return false;
            frame.rows = h_; frame.cols = w_; frame.nchannels = c_;
            frame.data = frame.storage.data();
        } else {
            char name[4096];
            std::snprintf(name, sizeof(name), path_.c_str(), (int)index_);
            if (index_ > 0 && std::string(name) == path_) return false;    // a single file, not a pattern
            frame = imread(name, false);
            if (frame.empty()) return false;
        }
        index_++;
        return true;
    }
}
