// Minimal stand-in for the cv::Mat arguments of the reference's interfaces
// (OpenCV is not a dependency of this build): a non-owning or owning 8-bit image
// with the member names the reference code reads (rows, cols, data, channels(),
// empty()), plus binary PGM / PPM and raw-frame file I/O.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mif
{
    struct Image
    {
        int rows = 0;
        int cols = 0;
        int nchannels = 1;
        unsigned char* data = nullptr;           // rows * cols * nchannels bytes; BGR order when 3 channels
        std::vector<unsigned char> storage;      // backs `data` when the image owns its pixels

        int channels() const { return nchannels; }
        bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }

        static Image wrap(unsigned char* pixels, int rows, int cols, int channels)
        {
            Image im;
            im.rows = rows; im.cols = cols; im.nchannels = channels; im.data = pixels;
            return im;
        }
    };

    // Reads a binary PGM (P5) or PPM (P6, converted from RGB to BGR order) file. With grayscale = true
    // a colour file is reduced to one channel with the BT.601 integer weights cv::imread uses
    // (IMREAD_GRAYSCALE). Returns an empty image on failure.
    Image imread(const std::string& path, bool grayscale);

    // Writes P5 / P6 (BGR converted back to RGB). Returns false on failure.
    bool imwrite(const std::string& path, const Image& image);

    // Sequence of raw frames, the stand-in for cv::VideoCapture: either a file of back-to-back
    // width*height*channels-byte frames described as "path:WxHxC", or a printf-style pattern of
    // PGM / PPM files ("frames/%04d.ppm", numbered from 0).
    class FrameReader
    {
    public:
        explicit FrameReader(const std::string& spec);
        bool isOpened() const { return opened_; }
        bool read(Image& frame);                 // false at the end of the sequence
        void seek(size_t index) { index_ = index; }      // the next read returns frame `index`
        size_t count(size_t limit) const;        // frames in the sequence, counted up to `limit`
    private:
        bool opened_ = false, raw_ = false;
        std::string path_;
        int w_ = 0, h_ = 0, c_ = 0;
        size_t index_ = 0;
    };
}
